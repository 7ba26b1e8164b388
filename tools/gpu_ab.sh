# A/B of library switches on a C3 slice and a C5 chunk: tools/gpu_ab.sh VAR=VALUE   (through gpurun)
V=${1:-HL_X=1}
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for v in A B A B; do
  case $v in A) E="HL_X=1";; B) E="$V";; esac
  echo "run $v ($E)" | tee -a gpurun_out/ab/ab.txt
  timeout -k 5 200 env $E python -u tools/slice_probe.py C3 0 3 2>&1 | grep --line-buffered -E "^rep|rror" | cut -c1-200 | sed -u "s/^/$v /" | tee -a gpurun_out/ab/ab.txt
done
timeout -k 10 500 python -u tools/c5_ab.py C5 26 60 - $V - $V 2>&1 | grep --line-buffered -E "rep|rror" | cut -c1-400 | tee -a gpurun_out/ab/ab.txt
