#!/bin/bash
# One round's measurement on the GPU box (run through gpurun from the repo root, each part fits one call):
#   tools/measure_round.sh <tag> c3      C3 profile passes (kernel stats, HBM traffic, SQ counters) + the bench lines of C3 and C2
#   tools/measure_round.sh <tag> c5      C5 profile passes + the bench line of C5 (full size, 8 chunks from chunk 26)
#   tools/measure_round.sh <tag> rest    the bench lines of C4s and C4 (full size, long reads)
# everything lands in gpurun_out/<tag>_*: copy what should be judged into profiles/
set -e
tag=${1:-rXX}; part=${2:-c3}
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
line() { wl=$1; shift; f=gpurun_out/${tag}_bench_$(echo $wl | tr A-Z a-z); python bench.py --workload $wl "$@" > $f.json 2> $f.err || echo "$wl BENCH FAILED"; tail -c 400 $f.json; }
case $part in
c3)   bash tools/profile_round.sh $tag C3 2>&1 | tail -5
      line C3 --steps 20 --warmup 5
      line C2 --steps 10 --warmup 3 ;;
c5)   bash tools/profile_round.sh $tag C5 2>&1 | tail -3
      line C5 --slice0 26 --steps 8 --warmup 1 --no-graph; rm -f /tmp/hlb/C5* ;;
rest) line C4s --steps 8 --warmup 1 --no-graph; rm -f /tmp/hlb/C4*
      line C4 --slice0 30 --steps 4 --warmup 1 --no-graph ;;
esac
