cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
timeout -k 10 600 python -u -m pytest tests/test_gpu_ava.py tests/test_gpu_short.py tests/test_gpu_workloads_oracle.py tests/test_gpu_stub.py tests/test_gpu_ungapped.py -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04k/small_tests.log | grep --line-buffered -E "passed|failed|FAILED|Error" | cut -c1-300
HL_CPU_BUDGET_S=1 timeout -k 10 400 python bench.py --workload C4s --steps 4 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r04k/c4s.json 2> gpurun_out/r04k/c4s.err || echo C4s FAILED
python -c "
import json; d=json.load(open('gpurun_out/r04k/c4s.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])"
timeout -k 5 200 python -u tools/slice_probe.py C3 0 3 2>&1 | grep --line-buffered -E "^rep|rror" | cut -c1-300
