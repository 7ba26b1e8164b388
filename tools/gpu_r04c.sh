set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout -k 10 300 python -u -m pytest "tests/test_gpu_ava.py::test_ava_noisy_reads" -v -p no:cacheprovider --tb=long 2>&1 | tee gpurun_out/r04c/noisy.log | tail -60 || true
timeout -k 10 1050 python -u -m pytest tests -m gpu -v -p no:cacheprovider --durations=30 --deselect tests/test_gpu_ava.py 2>&1 | tee gpurun_out/r04c/tests.log | grep --line-buffered -E "PASSED|FAILED|ERROR|passed|failed" || true
