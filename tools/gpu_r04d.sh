set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout -k 10 1100 python -u -m pytest "tests/test_gpu_ava.py::test_ava_noisy_reads" tests/test_gpu_c2_chunks_oracle.py tests/test_gpu_driver.py tests/test_gpu_full_c5.py tests/test_gpu_full_divergent.py tests/test_gpu_fullsize.py -v -s -p no:cacheprovider --durations=20 2>&1 | tee gpurun_out/r04d/tests.log | grep --line-buffered -E "PASSED|FAILED|ERROR|passed|failed|full size" || true
