# Step time of a C3 slice against the number of lanes (query batches in flight): LANES_LIST="1 2" tools/gpu_lanes.sh [workload] (through gpurun)
WL=${1:-C3}
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lanes
for n in ${LANES_LIST:-1 2 3 2 1}; do
  echo "lanes $n" | tee -a gpurun_out/lanes/lanes.txt
  timeout -k 5 240 env HLMI_LANES=$n python -u tools/slice_probe.py $WL 0 4 2>&1 | grep --line-buffered -E "^rep|rror|wall_s" | cut -c1-1500 | sed -u "s/^/L$n /" | tee -a gpurun_out/lanes/lanes.txt || exit 1
done
