#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv), with the ratios
that say what a kernel is bound by.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave
(MI355X_MICROARCH.md, 'rocprofv3 PMC slots'):
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES   (parked / issue-stalled / issuing)
usage: summarize_sq.py <out.json> <counter_collection.csv of pass 1> [<csv of pass 2> ...]"""
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
    out_path, csvs = sys.argv[1], sys.argv[2:]
    rows = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    for path in csvs:                       # one file per --pmc pass; a counter present in two passes keeps the first
        seen = defaultdict(lambda: defaultdict(float))
        ids = defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            seen[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ids[k].add(r.get("Dispatch_Id", ""))
        for k, c in seen.items():
            launches[k] = launches[k] or len(ids[k])
            for n, v in c.items():
                rows[k].setdefault(n, v)
    out = {}
    for k, c in rows.items():
        d = dict(launches=launches[k], **{n: v for n, v in sorted(c.items())})
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if n in c:
                    d["frac_" + n[3:].lower()] = round(c[n] / wc, 4)
        if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
            # share of the vector-issue cycles of the 1024 SIMDs (a wave64 VALU instruction holds its SIMD 4 cycles;
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            d["valu_issue_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
            if wc:
                d["waves_per_simd"] = round(wc * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 2)
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if n in c:
                    d[n[3:].lower() + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
        out[k] = d
    json.dump(out, open(out_path, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        print(k, {a: b for a, b in d.items() if a.startswith("frac_") or a.endswith("_per_wave") or a in ("launches", "valu_issue_frac", "waves_per_simd")})


if __name__ == "__main__":
    main()
