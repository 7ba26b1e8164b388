cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
for v in A B C A B C; do
  case $v in A) E="";; B) E="HLMI_LONG_MAIN_STREAM=1";; C) E="HLMI_LONG_LAST=1";; esac
  env $E python -u tools/slice_probe.py C3 0 3 2>&1 | grep -E "^rep|^\{\"align" | cut -c1-400 | sed "s/^/$v /" | tee -a gpurun_out/r04k/ab.txt
done
