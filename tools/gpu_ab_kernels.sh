# A/B of a library switch with the kernel timers of a C3 slice: tools/gpu_ab_kernels.sh VAR=VALUE [workload] (through gpurun)
V=${1:-HL_X=1}
WL=${2:-C3}
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/abk
for v in A B A B; do
  case $v in A) E="HL_X=1";; B) E="$V";; esac
  echo "run $v ($E)" | tee -a gpurun_out/abk/abk.txt
  timeout -k 5 240 env $E python -u tools/slice_probe.py $WL 0 3 2>&1 | cut -c1-1800 | sed -u "s/^/$v /" | tee -a gpurun_out/abk/abk.txt || exit 1
done
