set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout -k 10 1000 python -m pytest tests/test_gpu_ava.py tests/test_gpu_stub.py tests/test_gpu_short.py tests/test_gpu_workloads_oracle.py tests/test_gpu_c2_chunks_oracle.py -x -q -s > gpurun_out/r04b/tests.log 2>&1 || echo "TEST FAILED" >> gpurun_out/r04b/tests.log
tail -30 gpurun_out/r04b/tests.log
