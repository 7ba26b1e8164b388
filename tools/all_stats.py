import json, os, sys, tempfile
sys.path.insert(0, os.getcwd())
from hylight_amd import api, workloads as W
from hylight_amd.stage import StageRunner
cfg = W.config("C3", 1.0)
work = os.environ.get("HL_BENCH_DIR") or tempfile.mkdtemp(prefix="hl_probe_")
fa = os.path.join(work, cfg["name"] + ".fa")
if not os.path.exists(fa): W.make_long(cfg, fa)
api.init(0, 0)
r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True); r.prepare()
for k in range(2):
    n = r.run(os.path.join(work, "probe.paf"), share=(0, 8), **cfg["stage"])
st = api.last_stats()
print(json.dumps({k: v for k, v in sorted(st.items()) if not k.startswith("kernel_launches")}))
