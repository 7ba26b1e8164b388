"""Build libhylight_mi.so in-tree with hipcc for gfx950 (no torch in the link; C ABI only).

    python -m hylight_amd.build [--force]

Object files are cached under hylight_amd/csrc/build/ keyed by source mtime.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhylight_mi.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]
# A kernel that runs out of registers still builds: it spills to scratch and runs several times slower (round 5: a dozen lines
# more in classify_kernel<1>, pinned to 128 registers for its occupancy, spilled 305 of them and cost the C3 step 170 ms).  The
# build reads hipcc's resource remarks and refuses a kernel with more spilled vector registers than this.
MAX_VGPR_SPILL = 16
if os.environ.get("HLMI_INSTRUMENT"):            # kernel phase counters / self-checks (tuning builds only, never benched)
    FLAGS.append("-DHLMI_INSTRUMENT")


def resource_remarks(stderr: str):
    """hipcc's -Rpass-analysis=kernel-resource-usage output -> ([(kernel, spilled vector registers)] above MAX_VGPR_SPILL,
    the other lines: warnings and errors worth showing)."""
    name, spilled, other = "?", [], []
    for line in stderr.splitlines():
        if "-Rpass-analysis=kernel-resource-usage" in line or "remark:" in line:
            if "Function Name:" in line:
                name = line.split("Function Name:")[1].split("[")[0].strip()
            elif "VGPRs Spill:" in line:
                n = int(line.split("VGPRs Spill:")[1].split("[")[0])
                if n > MAX_VGPR_SPILL:
                    spilled.append((name, n))
        elif not re.match(r"^\s*(\d+\s*)?\|", line) and not re.match(r"^\d+ (remark|warning)s? generated", line):
            other.append(line)                 # (not the source excerpt under a remark)
    return spilled, other


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
        if src.endswith(".hip"):
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print("[build]", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        spilled, other = resource_remarks(r.stderr)
        if spilled:
            os.remove(obj)
            raise RuntimeError(f"{os.path.basename(src)}: kernels spill vector registers to scratch: " +
                               ", ".join(f"{k} ({n})" for k, n in spilled))
        if "\n".join(other).strip() and verbose:
            print("\n".join(other), file=sys.stderr)

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
