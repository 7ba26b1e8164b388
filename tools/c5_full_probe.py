#!/usr/bin/env python3
"""One --nsplit chunk of the FULL C5 (500 000 reads) with every library statistic printed (needs the GPU)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hylight_amd import api, workloads as W
from hylight_amd.stage import StageRunner
import torch
torch.cuda.set_device(0)
api.init(0, 0)
cfg = W.config(sys.argv[1] if len(sys.argv) > 1 else "C5")
d = tempfile.mkdtemp(prefix="hl_probe_")
fa = os.path.join(d, "r.fa")
t = time.time(); n, bases, _ = W.make_long(cfg, fa); print("simulate", n, bases, round(time.time() - t, 1), flush=True)
t = time.time(); r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True); print("open", round(time.time() - t, 1), r.job.num_queries, r.job.num_chunks, flush=True)
t = time.time(); r.prepare(); print("sketch", round(time.time() - t, 2), flush=True)
share = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (17, 60)
t = time.time(); rows = r.run(os.path.join(d, "o.paf"), share=share, **cfg["stage"]); print("run", round(time.time() - t, 1), rows, flush=True)
print(json.dumps({k: round(v, 4) for k, v in sorted(api.last_stats().items())}))
r.close()
os.remove(fa)
