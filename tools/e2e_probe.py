#!/usr/bin/env python3
"""End-to-end time of the drop-in call split_reads2(fa, fa, nsplit, ...) on the C2 workload: FASTA parsing, upload, name
ranks, the stage pass and the output file - what HyLight.py would wait for.  usage: python tools/e2e_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hylight_amd import api  # noqa: E402

wl = bench.WORKLOADS["C2"]
work = os.environ.get("HL_BENCH_DIR", "/tmp/hlb")
os.makedirs(work, exist_ok=True)
fa = os.path.join(work, "C2.fa")
if not os.path.exists(fa):
    bench.make_workload("C2", fa)
api.init(0, 0)
for it in range(3):
    t0 = time.time()
    job = api.Job(fa, fa, wl["nsplit"], True)
    t1 = time.time()
    job.close()
    t2 = time.time()
    api.split_reads2(fa, fa, wl["nsplit"], work, os.path.join(work, "e2e.paf"), threads=8, len_over=6000, mc=2, iden=0.95, long=True)
    t3 = time.time()
    print(it, "job_open %.1f ms, split_reads2 end to end %.1f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
