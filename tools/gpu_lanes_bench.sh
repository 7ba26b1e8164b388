# bench.py of a workload against the number of lanes: tools/gpu_lanes_bench.sh <workload> "<lanes list>" [bench args] (through gpurun)
WL=${1:-C4s}; LIST=${2:-"2 3 4"}; shift; shift
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 HL_BENCH_DIR=/tmp/hlb && mkdir -p $HL_BENCH_DIR && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lanes_bench
for n in $LIST; do
  f=gpurun_out/lanes_bench/${WL}_lanes$n.json
  timeout -k 10 400 env HLMI_LANES=$n python bench.py --workload $WL --no-cpu-baseline --no-graph "$@" > $f 2> $f.err || { echo "lanes $n failed"; tail -3 $f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
print("lanes $n:", round(d["ms_per_step"],1), "ms per step", [round(x) for x in d["step_ms"]])
PY
done
