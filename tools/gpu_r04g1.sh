set -e
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04p C5 2>&1 | tail -4
