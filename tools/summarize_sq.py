#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv), with the ratios
that say what a kernel is bound by.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave
(MI355X_MICROARCH.md, 'rocprofv3 PMC slots'):
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES   (parked / issue-stalled / issuing)

valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): a wave64 VALU instruction occupies its
SIMD-32 for 2 cycles (MI355X_MICROARCH.md, wave / SIMD).  Calibration (tools/valu_calib.hip: nothing but independent
v_add_u32 at 8 resident waves per SIMD; profiles/r04_valu_calibration.json): that kernel reads 0.893 - the issue rate a
saturated SIMD really sustains is 2.24 cycles per instruction - so `valu_issue_vs_saturation` = valu_issue_frac / 0.893 is
the share of the attainable rate, 1.0 on the calibration kernel.  (Rounds 1-3 charged SQ_ACTIVE_INST_VALU x 4 cycles:
that counter equals the instruction count on the calibration kernel, and the factor 4 made a saturated SIMD read 1.79.)
`wave_cycles_per_simd_cycle` = SQ_WAVE_CYCLES x 4 / SIMD cycles is NOT calibrated as an occupancy: the calibration kernel
holds 8 waves per SIMD by construction and reads 4.74.
usage: summarize_sq.py <out.json> <counter_collection.csv of pass 1> [<csv of pass 2> ...]"""
VALU_CYCLES = 2.0            # MI355X_MICROARCH.md: cycles a wave64 VALU instruction holds its SIMD-32
VALU_SATURATION = 0.893      # valu_issue_frac of tools/valu_calib.hip (profiles/r04_valu_calibration.json)
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
        if c.get("GRBM_GUI_ACTIVE") and ("SQ_INSTS_VALU" in c or "SQ_ACTIVE_INST_VALU" in c):
            # share of the vector-issue cycles of the 1024 SIMDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  The two passes
            # of a profile see the same launches; SQ_ACTIVE_INST_VALU (pass 1) equals the instruction count (calibration).
            insts = c.get("SQ_INSTS_VALU", c.get("SQ_ACTIVE_INST_VALU"))
            simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
            d["valu_issue_frac"] = round(insts * VALU_CYCLES / simd_cycles, 4)
            d["valu_issue_vs_saturation"] = round(insts * VALU_CYCLES / simd_cycles / VALU_SATURATION, 4)
            if wc:
                d["wave_cycles_per_simd_cycle"] = round(wc * 4.0 / simd_cycles, 2)
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if n in c:
                    d[n[3:].lower() + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
        out[k] = d
    json.dump(out, open(out_path, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        print(k, {a: b for a, b in d.items() if a.startswith("frac_") or a.endswith("_per_wave") or a in ("launches", "valu_issue_frac", "valu_issue_vs_saturation", "wave_cycles_per_simd_cycle")})


if __name__ == "__main__":
    main()
