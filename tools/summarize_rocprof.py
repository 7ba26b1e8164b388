#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a short table (kernel names cut
to their function name) for profiles/.   usage: summarize_rocprof.py <kernel_stats.csv> [top_n]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(radix_sort_onesweep_\w+|partition_impl|scan_impl|lookback_scan_\w+|__amd_rocclr_\w+)", name)
    if "rocprim" in name and m:
        name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "", name)
        # key / value types, and the block shape when the sort was given one (dev_prims.hip)
        kv = re.search(r"wrapped_radix_sort_onesweep_config<(?:default_config|radix_sort_onesweep_config<.*?\)\d+>), ([^,]+), ([^>]+)>", name)
        shape = re.search(r"radix_sort_onesweep_config<kernel_config<(\d+)u, (\d+)u, \d+u>, kernel_config<(\d+)u, (\d+)u, \d+u>", name)
        tag = f"<{kv.group(1)},{kv.group(2)}>" if kv else ""
        if shape:
            tag += f"[hist {shape.group(1)}x{shape.group(2)}, sort {shape.group(3)}x{shape.group(4)}]"
        return "rocprim::" + m.group(1) + tag
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}")
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:72s} {c:7d} {ns / 1e6:10.2f} {ns / c / 1e3:10.1f} {100 * ns / tot:6.2f}")
    print(f"{'TOTAL':72s} {sum(v[0] for v in agg.values()):7d} {tot / 1e6:10.2f}")


if __name__ == "__main__":
    main()
