# round-4 measurement, part 1: C3 profile passes + the bench lines of C3 and C2
set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r04p C3 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/r04p_bench_c3.json 2> gpurun_out/r04p_bench_c3.err || echo "C3 BENCH FAILED"
tail -c 600 gpurun_out/r04p_bench_c3.json
python bench.py --workload C2 --steps 10 --warmup 3 > gpurun_out/r04p_bench_c2.json 2> gpurun_out/r04p_bench_c2.err || echo "C2 BENCH FAILED"
tail -c 400 gpurun_out/r04p_bench_c2.json
