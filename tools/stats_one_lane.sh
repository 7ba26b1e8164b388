# kernel stats of the C3 bench step with ONE lane (every kernel alone on the card): tools/stats_one_lane.sh <tag> (through gpurun)
tag=${1:-rXX}
out=$PWD/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb HLMI_LANES=1 && mkdir -p $HL_BENCH_DIR && cd "$OLDPWD"
rm -rf "$out/${tag}_stats1"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats1" -o p --output-format csv -- python bench.py --workload C3 --steps 2 --warmup 1 --no-cpu-baseline --no-graph > "$out/${tag}_stats1.log" 2>&1
python tools/summarize_rocprof.py "$out/${tag}_stats1/p_kernel_stats.csv" > "$out/${tag}_c3_kernel_stats_one_lane.txt"
find "$out/${tag}_stats1" -name "*.csv" -size +1M -delete
head -24 "$out/${tag}_c3_kernel_stats_one_lane.txt"
