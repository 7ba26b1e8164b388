#!/usr/bin/env python3
"""Per-kernel launch histogram of a rocprofv3 --kernel-trace CSV: for every kernel name (shortened) the calls, the total
time and the launches grouped by grid size.  usage: trace_hist.py <p_kernel_trace.csv> [name substring ...]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
by = collections.defaultdict(list)
for r in rows:
    nm = re.sub(r"ROCPRIM_400200_NS::detail::|rocprim::|hlmi::|unsigned ", "", r["Kernel_Name"])[:170]
    by[nm].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
tot = sum(d for v in by.values() for d, _ in v)
for nm, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    if want and not any(w in nm for w in want): continue
    t = sum(d for d, _ in v)
    print(f"{nm[:170]:170s} calls {len(v):6d} total {t/1e6:9.2f} ms  {100*t/tot:5.2f} %")
    if want:
        g = collections.defaultdict(list)
        for d, gs in v: g[gs].append(d)
        for gs, ds in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:12]:
            print(f"      grid {gs:12d}: {len(ds):5d} calls, {sum(ds)/1e6:8.2f} ms, avg {sum(ds)/len(ds)/1e3:9.1f} us")
