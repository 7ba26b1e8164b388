set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/r04c/tests.log 2>&1 || echo "TEST FAILED" >> gpurun_out/r04c/tests.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04c/tests.log | tail -40
