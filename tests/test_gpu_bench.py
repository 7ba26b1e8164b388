"""GPU: the contract of bench.py's one JSON line (a tenth of C2 so that it takes seconds): the fields the driver reads, the
roofline block of the dominant kernel and the CPU baseline beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_every_contracted_field(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "C2", "--scale", "0.1"],
                       env=dict(env, HL_BENCH_DIR=str(tmp_path), HL_CPU_BUDGET_S="20"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "long-read all-vs-all overlaps/sec" and d["unit"] == "overlaps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - d["config"]["overlaps_out"] / (3 * d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"] and len(d["step_ms"]) == 3
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "candidate_rows_per_s"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["candidate_rows_per_s"] > 0
    # whole --nsplit chunks, and the GPU's rows for the first of them held against the oracle's inside the bench leg
    assert cb["chunks"] and "WHOLE" in cb["sample"]
    if cb["kind"] == "port":
        assert cb["parity"]["gpu_rows_identical"] is True and cb["parity"]["gpu_worker_identical"] is True
    assert d["graph_build_s"] is not None and d["graph_rows_in"] > 0
