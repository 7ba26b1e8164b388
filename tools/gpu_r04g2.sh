set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --workload C5 --slice0 26 --steps 8 --warmup 1 --no-graph > gpurun_out/r04p_bench_c5.json 2> gpurun_out/r04p_bench_c5.err || echo "C5 BENCH FAILED"
tail -c 300 gpurun_out/r04p_bench_c5.json; rm -f /tmp/hlb/C5*
python bench.py --workload C4 --slice0 30 --steps 4 --warmup 1 --no-graph > gpurun_out/r04p_bench_c4.json 2> gpurun_out/r04p_bench_c4.err || echo "C4 BENCH FAILED"
tail -c 300 gpurun_out/r04p_bench_c4.json
