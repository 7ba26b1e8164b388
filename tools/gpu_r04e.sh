set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout -k 10 400 python -u tools/c5_full_probe.py C5 40 60 2>&1 | tee gpurun_out/r04e/c5_chunk40_b.txt | cut -c1-300
timeout -k 10 600 python -u -m pytest tests/test_gpu_workloads_oracle.py::test_c5_depth_sample_matches_the_oracle tests/test_gpu_stub.py tests/test_gpu_full_c5.py::test_c5_is_dp_bound_and_rows_are_valid -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/r04e/tests.log | tail -5 | cut -c1-600
