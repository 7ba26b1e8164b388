#!/usr/bin/env python3
"""A/B of library switches on one --nsplit chunk of the FULL C5 (needs the GPU): the read set is simulated and sketched once,
then every variant (NAME=VALUE,... or '-') runs the same chunk twice.  usage: c5_ab.py [C5|C4] slice of variant..."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hylight_amd import api, workloads as W
from hylight_amd.stage import StageRunner
import torch
torch.cuda.set_device(0)
api.init(0, 0)
cfg = W.config(sys.argv[1])
share = (int(sys.argv[2]), int(sys.argv[3]))
d = tempfile.mkdtemp(prefix="hl_probe_")
fa = os.path.join(d, "r.fa")
t = time.time(); n, bases, _ = W.make_long(cfg, fa); print("simulate", n, bases, round(time.time() - t, 1), flush=True)
r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
r.prepare()
print("ready", flush=True)
for var in sys.argv[4:] or ["-"]:
    keys = []
    if var != "-":
        for kv in var.split(","):
            k, v = kv.split("="); os.environ[k] = v; keys.append(k)
    for rep in range(2):
        t = time.time(); rows = r.run(os.path.join(d, "o.paf"), share=share, **cfg["stage"]); dt = time.time() - t
        st = api.last_stats()
        km = {k[len("kernel_ms."):]: round(v, 1) for k, v in st.items() if k.startswith("kernel_ms.") and v >= 20}
        print(f"{var} rep {rep}: {dt:.2f} s, {rows} rows, t_ava {st['t_ava_s']:.2f}", json.dumps(dict(sorted(km.items(), key=lambda kv: -kv[1]))), flush=True)
    for k in keys: del os.environ[k]
r.close()
os.remove(fa)
