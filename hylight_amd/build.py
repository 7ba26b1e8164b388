"""Build libhylight_mi.so in-tree with hipcc for gfx950 (no torch in the link; C ABI only).

    python -m hylight_amd.build [--force]

Object files are cached under hylight_amd/csrc/build/ keyed by source mtime.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhylight_mi.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]
if os.environ.get("HLMI_INSTRUMENT"):            # kernel phase counters / self-checks (tuning builds only, never benched)
    FLAGS.append("-DHLMI_INSTRUMENT")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "hylight_mi.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = True) -> str:
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    hdr_t = _newest_header()
    todo, objs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print("[build]", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=min(6, len(todo))) as ex:
            list(ex.map(cc, todo))
    if todo or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print("[build] linked", OUT, flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
