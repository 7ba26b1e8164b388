cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
timeout -k 10 300 python bench.py --workload C4s --steps 4 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r04k/c4s.json 2> gpurun_out/r04k/c4s.err || echo C4s FAILED
python -c "
import json; d=json.load(open('gpurun_out/r04k/c4s.json')); print(d['ms_per_step'], d['value'], d['stage_seconds'])"
f=/tmp/hlb/out.paf.slice0.call1
if [ -s $f ]; then wc -l < $f; cut -f12 < $f | uniq -c | sort -nr | head -5; fi

timeout -k 10 600 python -u -m pytest tests/test_gpu_filters.py tests/test_gpu_tiebreak.py tests/test_gpu_driver.py tests/test_gpu_properties.py tests/test_gpu_short.py -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04k/small_tests.log | grep --line-buffered -E "passed|failed|FAILED|Error" | cut -c1-300
