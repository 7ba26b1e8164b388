#!/bin/bash
# kernel stats only: tools/stats_only.sh <tag> [workload]
set -u
tag=${1:-rXX}; WL=${2:-C3}; wl=$(echo $WL | tr A-Z a-z)
out=$PWD/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd "$OLDPWD"
rm -rf "$out/${tag}_stats"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats" -o p --output-format csv -- python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-graph > "$out/${tag}_stats.log" 2>&1
python tools/summarize_rocprof.py "$out/${tag}_stats/p_kernel_stats.csv" > "$out/${tag}_${wl}_kernel_stats.txt"
find "$out" -name "*.csv" -size +1M -delete
head -30 "$out/${tag}_${wl}_kernel_stats.txt"
