#!/bin/bash
# Collect the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (two --pmc passes), SQ counters (two --pmc passes).
# usage: tools/profile_round.sh <tag> [workload=C3]      -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
set -u
tag=${1:-rXX}
out=$PWD/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd "$OLDPWD"
WL=${2:-C3}
wl=$(echo $WL | tr A-Z a-z)
B="python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
run() { name=$1; shift; rm -rf "$out/$name"; timeout -k 10 500 rocprofv3 "$@" -d "$out/$name" -o p --output-format csv -- $B > "$out/$name.log" 2>&1; }
run ${tag}_stats --kernel-trace --stats && python tools/summarize_rocprof.py "$out/${tag}_stats/p_kernel_stats.csv" > "$out/${tag}_${wl}_kernel_stats.txt"
run ${tag}_fetch --pmc FETCH_SIZE || true
run ${tag}_write --pmc WRITE_SIZE || true
python tools/summarize_pmc.py "$out/${tag}_fetch/p_counter_collection.csv" "$out/${tag}_write/p_counter_collection.csv" "$out/${tag}_${wl}_pmc_traffic.json" > "$out/${tag}_${wl}_pmc_traffic.txt" 2>&1
run ${tag}_sq1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run ${tag}_sq2 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python tools/summarize_sq.py "$out/${tag}_${wl}_sq_counters.json" "$out/${tag}_sq1/p_counter_collection.csv" "$out/${tag}_sq2/p_counter_collection.csv" > "$out/${tag}_${wl}_sq_counters.txt"
sha256sum hylight_amd/libhylight_mi.so | cut -c1-16 > "$out/${tag}_${wl}_lib_sha.txt"
find "$out" -name "*.csv" -size +1M -delete
ls "$out" | grep "^${tag}_" | head -30
