#!/usr/bin/env python3
"""Stage statistics (host phases, kernel times) of ONE rank's share of an N-rank run, on one GPU.
usage: HLMI_HOST_TIMERS=1 python tools/rank_stats.py [N] [rank] [workload]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from hylight_amd import api  # noqa: E402
from hylight_amd.stage import StageRunner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 0
name = sys.argv[3] if len(sys.argv) > 3 else "C2"
wl = bench.WORKLOADS[name]
work = os.environ.get("HL_BENCH_DIR", "/tmp/hlb")
os.makedirs(work, exist_ok=True)
fa = os.path.join(work, name + ".fa")
if not os.path.exists(fa):
    bench.make_workload(name, fa)
api.init(0, 0)
runner = StageRunner(fa, fa, wl["nsplit"], long_mode=True)
stage = wl.get("stage", bench.STAGE)
out = os.path.join(work, "probe.paf")
for _ in range(2):
    runner._install_sketch()
    torch.cuda.synchronize()
    t0 = time.time()
    runner.job.run(r, n, stage["len_over"], stage["mc"], stage["iden"], out)
    torch.cuda.synchronize()
    dt = time.time() - t0
st = api.last_stats()
print(json.dumps({"n": n, "rank": r, "seconds": round(dt, 4),
                  "stats": {k: round(v, 4) for k, v in sorted(st.items()) if k.startswith(("host_s", "kernel_ms", "t_"))}}))
