# the opt-in complete C5 pass (60 chunks on one card, then the two shares of a 2-rank job): through gpurun, ~8 minutes
set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_FULL_PASS=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -u -m pytest tests/test_gpu_fullsize.py::test_c5_full_size_complete_pass "tests/test_gpu_short.py::test_small_groups_chain_the_same_in_both_kernels" -x -v -s -p no:cacheprovider 2>&1 | tee gpurun_out/r04_c5_complete_pass.log | grep --line-buffered -E "full size|share|PASSED|FAILED|passed|failed|Error" || true
