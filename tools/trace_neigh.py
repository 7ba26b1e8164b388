#!/usr/bin/env python3
"""Neighbours of the slow launches of one kernel in a rocprofv3 kernel trace.  usage: trace_neigh.py <csv> <substring> <min_us>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
sub, min_us = sys.argv[2], float(sys.argv[3])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"ROCPRIM_400200_NS::detail::|rocprim::|hlmi::|unsigned |\(anonymous namespace\)::", "", r["Kernel_Name"])[:130], r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows))
t0 = ev[0][0]
for i, e in enumerate(ev):
    if sub in e[2] and (e[1] - e[0]) / 1e3 >= min_us:
        print(f"--- at {(e[0]-t0)/1e6:9.2f} ms: {e[2]} dur {(e[1]-e[0])/1e3:.0f} us queue {e[3]} grid {e[4]}")
        for j in range(max(0, i - 3), min(len(ev), i + 4)):
            f = ev[j]
            mark = "*" if j == i else " "
            print(f"   {mark} {(f[0]-t0)/1e6:9.2f} +{(f[1]-f[0])/1e3:8.0f} us q{f[3]} g{f[4]:>10} {f[2]}")
        over = [f for f in ev if f is not e and f[0] < e[1] and f[1] > e[0]]
        for f in over[:4]: print(f"     overlaps: {(f[0]-t0)/1e6:9.2f} +{(f[1]-f[0])/1e3:8.0f} us q{f[3]} {f[2]}")
