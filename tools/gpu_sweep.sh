# step time of a C3 slice under a list of settings: tools/gpu_sweep.sh "VAR=a" "VAR=b" ... (through gpurun; "HL_X=1" = the defaults)
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sweep
for E in "$@"; do
  echo "setting $E" | tee -a gpurun_out/sweep/sweep.txt
  timeout -k 5 240 env $E python -u tools/slice_probe.py ${WL:-C3} 0 4 2>&1 | grep --line-buffered -E "^rep|rror|wall_s" | cut -c1-600 | sed -u "s/^/[$E] /" | tee -a gpurun_out/sweep/sweep.txt || exit 1
done
