import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hylight_amd import api
from hylight_amd.stage import StageRunner
wl = bench.WORKLOADS["C2"]
work = "/tmp/hlb"; os.makedirs(work, exist_ok=True)
fa = os.path.join(work, "C2.fa")
if not os.path.exists(fa): bench.make_workload("C2", fa)
api.init(0, 0)
r = StageRunner(fa, fa, wl["nsplit"], long_mode=True)
out = os.path.join(work, "o.paf")
r.run(out, 6000, 2, 0.95)
for it in range(3):
    t0 = time.time()
    api.miniasm(out, fa, os.path.join(work, "c.gfa"), bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1)
    dt = time.time() - t0
    st = api.last_stats()
    print(it, round(dt * 1e3, 1), {k: round(v * 1e3, 1) for k, v in st.items() if k.startswith("t_graph")})
