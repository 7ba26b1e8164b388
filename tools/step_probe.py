#!/usr/bin/env python3
"""Where one bench step goes on the host side: query sketch, hlmi_job_run (its own clock, and the same with the
clean-up of its locals), statistics read-back.  usage: python tools/step_probe.py"""
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
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    r._install_sketch(); torch.cuda.synchronize(); t1 = time.time()
    r.job.run(0, 1, 6000, 2, 0.95, out); torch.cuda.synchronize(); t2 = time.time()
    st = api.last_stats(); t3 = time.time()
    print(it, "sketch %.1f run %.1f (t_total %.1f, with cleanup %.1f) stats %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, st["t_total_s"]*1e3, st.get("t_with_cleanup_s",0)*1e3, (t3-t2)*1e3))
