#!/usr/bin/env python3
"""Single-GPU estimate of the N-GPU step: times what ONE rank of an N-rank run does (its 1/N slice of the
sketch, its chunks i % N == r of the stage) for N = 1, 2, 4, 8 and every r.  The RCCL all-gather of the sketches
(16 B x 2e7 minimizers on C2) and the rank-0 merge are not included; bench.py --gpus N measures the real thing.
usage: python tools/scale_probe.py [workload]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
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
runner.run(out, **stage)                       # warm-up (pool, caches)
res = {}
for n in (1, 2, 4, 8):
    per_rank = []
    for r in range(n):
        for _ in range(2):                         # the first pass of a share sizes the device pool (like a bench warm-up)
            runner._install_sketch()
            torch.cuda.synchronize()
            t0 = time.time()
            runner.job.run(r, n, stage["len_over"], stage["mc"], stage["iden"], out)
            torch.cuda.synchronize()
            dt = time.time() - t0
        per_rank.append(dt)
    res[n] = dict(max_rank_s=round(max(per_rank), 4), mean_rank_s=round(sum(per_rank) / n, 4),
                  efficiency_vs_1=None if n == 1 else round(res[1]["max_rank_s"] / (n * max(per_rank)), 3))
print(json.dumps(res))
