cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
timeout -k 10 400 python -u -m pytest tests/test_gpu_short.py tests/test_gpu_ava.py tests/test_gpu_stub.py tests/test_gpu_workloads_oracle.py::test_c4_short_calls_sample_matches_the_oracle -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04k/small_tests.log | grep --line-buffered -E "passed|failed|FAILED|Error" | cut -c1-300
timeout -k 10 300 python bench.py --workload C4s --steps 4 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r04k/c4s.json 2> gpurun_out/r04k/c4s.err || echo C4s FAILED
python -c "
import json; d=json.load(open('gpurun_out/r04k/c4s.json')); print(d['ms_per_step'], d['value'], d['stage_seconds']['t_ava_s'])"
