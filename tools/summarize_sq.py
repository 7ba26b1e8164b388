#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv), with the ratios
that say what a kernel is bound by.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave
(MI355X_MICROARCH.md, 'rocprofv3 PMC slots'):
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES   (parked / issue-stalled / issuing)
usage: summarize_sq.py <counter_collection.csv> <out.json> [kernel-name-substring ...]"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if "rocprim" in name:
        m = re.search(r"(radix_sort_onesweep_\w+|partition_impl|scan_impl)", name)
        return "rocprim::" + (m.group(1) if m else "other")
    return re.sub(r"^void ", "", name).split("(")[0][:60]


def main():
    rows = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id", ""))
    want = sys.argv[3:]
    out = {}
    for k, c in rows.items():
        if want and not any(w in k for w in want):
            continue
        d = dict(launches=len(launches[k]), **{n: v for n, v in sorted(c.items())})
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if n in c:
                    d["frac_" + n[3:].lower()] = round(c[n] / wc, 4)
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if n in c:
                    d[n[3:].lower() + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
        out[k] = d
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        print(k, {a: b for a, b in d.items() if a.startswith("frac_") or a.endswith("_per_wave") or a == "launches"})


if __name__ == "__main__":
    main()
