#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads, so the read side is doubled (calibration point: encode_kernel reads 1 B/base
and reports 0.5 B/base).  usage: summarize_pmc.py <FETCH_csv> <WRITE_csv> <out.json>"""
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


def load(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    f, fc = load(sys.argv[1])
    w, wc = load(sys.argv[2])
    out = {}
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
        n = max(fc[k], 1)
        rd, wr = 2 * f[k] * 1024, w.get(k, 0.0) * 1024
        out[k] = dict(launches=n, read_bytes_per_launch=rd / n, write_bytes_per_launch=wr / n,
                      hbm_bytes_per_launch=(rd + wr) / n, hbm_bytes_total=rd + wr)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:16]:
        print(f"{k:46s} launches {v['launches']:5d}  read {v['read_bytes_per_launch']/1e6:10.1f} MB  write "
              f"{v['write_bytes_per_launch']/1e6:10.1f} MB  per launch;  total {v['hbm_bytes_total']/1e9:8.2f} GB")


if __name__ == "__main__":
    main()
