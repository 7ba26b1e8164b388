#!/usr/bin/env python3
"""Kernel timers of one bench step, repeated: `python tools/slice_probe.py [workload=C3] [slice=0] [reps=3] [scale=1]`.
Prints the per-kernel HIP-event times of the last repetition and the wall time of each (tuning aid)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hylight_amd import api, workloads as W   # noqa: E402
from hylight_amd.stage import StageRunner     # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
sl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
slices = {"C3": 8, "C2": 1}.get(wl, 8)
cfg = W.config(wl, scale)
work = os.environ.get("HL_BENCH_DIR") or tempfile.mkdtemp(prefix="hl_probe_")
fa = os.path.join(work, cfg["name"] + ".fa")
if not os.path.exists(fa):
    W.make_long(cfg, fa)
api.init(0, 0)
r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
r.prepare()
for k in range(reps):
    t = time.time()
    n = r.run(os.path.join(work, "probe.paf"), share=(sl, slices), **cfg["stage"])
    dt = time.time() - t
    st = api.last_stats()
    print(f"rep {k}: {dt * 1e3:.1f} ms, {n} rows, t_ava {st['t_ava_s']*1e3:.1f} t_filter {st['t_filter_s']*1e3:.1f}", flush=True)
kms = {k.split(".", 1)[1]: round(v, 2) for k, v in st.items() if k.startswith("kernel_ms.")}
print(json.dumps(dict(sorted(kms.items(), key=lambda kv: -kv[1]))))
print(json.dumps({k: v for k, v in st.items() if k.startswith("host_s.") or k.startswith("wall_s.") or k.startswith("chain_") or k in ("anchors", "align_tasks", "align_tasks_dp", "pieces", "ava_lanes", "lanes_fit", "hbm_peak_in_use_gb")}))
r.close()
