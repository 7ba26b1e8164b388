cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
for v in A B A B; do
  case $v in A) E="HL_X=1";; B) E="HLMI_NO_PIPELINE=1";; esac
  echo "run $v" | tee -a gpurun_out/r04k/ab.txt
  timeout -k 5 200 env $E python -u tools/slice_probe.py C3 0 3 2>&1 | grep --line-buffered -E "^rep|Error|error" | cut -c1-400 | sed -u "s/^/$v /" | tee -a gpurun_out/r04k/ab.txt || exit 1
done
