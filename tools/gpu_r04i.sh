set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
HL_BENCH_STATS=1 python -u tools/all_stats.py > gpurun_out/r04i/c3_all_stats.txt 2>&1 || true
tail -c 5000 gpurun_out/r04i/c3_all_stats.txt | python -c "
import sys,json
t=sys.stdin.read(); d=json.loads(t[t.index('{\"align_bases'):].split('\n')[0])
print({k[10:]:round(v,1) for k,v in d.items() if k.startswith('kernel_ms.')})
print({k:v for k,v in d.items() if k in ('t_total_s','t_ava_s','rows_out','ava_rows','align_tasks_long')})
"
timeout -k 10 900 python -u -m pytest tests/test_gpu_ava.py tests/test_gpu_stub.py tests/test_gpu_workloads_oracle.py tests/test_gpu_ungapped.py tests/test_gpu_short.py -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04i/tests.log | tail -5 | cut -c1-600
