# round-4 measurement, part 2: C5 profile passes, then the bench lines of C5 (full size), C4s and C4 (full size, long reads)
set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r04p C5 2>&1 | tail -3
cp gpurun_out/r04p_c5_pmc_traffic.json gpurun_out/r04p_c5_sq_counters.json profiles/ 2>/dev/null || true
python bench.py --workload C5 --slice0 26 --steps 8 --warmup 1 --no-graph > gpurun_out/r04p_bench_c5.json 2> gpurun_out/r04p_bench_c5.err || echo "C5 BENCH FAILED"
tail -c 400 gpurun_out/r04p_bench_c5.json; rm -f /tmp/hlb/C5*
python bench.py --workload C4s --steps 8 --warmup 1 --no-graph > gpurun_out/r04p_bench_c4s.json 2> gpurun_out/r04p_bench_c4s.err || echo "C4s BENCH FAILED"
tail -c 400 gpurun_out/r04p_bench_c4s.json; rm -f /tmp/hlb/C4*
