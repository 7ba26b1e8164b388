# the whole GPU suite with progress lines (a silent run is taken to be hung after 7 minutes)
set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1150 python -u -m pytest tests -m gpu -v -p no:cacheprovider --durations=30 2>&1 | tee gpurun_out/suite/tests.log | grep --line-buffered -E "PASSED|FAILED|ERROR|passed|failed|SKIPPED" || true
