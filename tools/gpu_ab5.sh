cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1 && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04e
python - <<'PY'
import json, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from hylight_amd import api, workloads as W
from hylight_amd.stage import StageRunner
import torch
torch.cuda.set_device(0); api.init(0, 0)
cfg = W.config("C5"); d = tempfile.mkdtemp(prefix="hl_probe_"); fa = os.path.join(d, "r.fa")
W.make_long(cfg, fa)
r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True); r.prepare()
for env in ("", "1", "", "1"):
    if env: os.environ["HLMI_LONG_MAIN_STREAM"] = "1"
    else: os.environ.pop("HLMI_LONG_MAIN_STREAM", None)
    t = time.time(); rows = r.run(os.path.join(d, "o.paf"), share=(40, 60), **cfg["stage"]); dt = time.time() - t
    st = api.last_stats()
    print("main_stream" if env else "side_stream", round(dt, 2), {k[10:]: round(v) for k, v in st.items() if k.startswith("kernel_ms.") and v > 150}, flush=True)
r.close()
PY
