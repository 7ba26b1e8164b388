set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout -k 10 400 python -u tools/c5_full_probe.py C5 40 60 2>&1 | tee gpurun_out/r04e/c5_chunk40_c.txt | cut -c1-200
timeout -k 10 900 python -u -m pytest tests/test_gpu_ava.py tests/test_gpu_stub.py tests/test_gpu_workloads_oracle.py tests/test_gpu_c2_chunks_oracle.py -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04e/tests.log | tail -5 | cut -c1-600
