#!/usr/bin/env python3
"""GPU idle time between the kernels of a frame, from a rocprofv3 --kernel-trace CSV of tools/bench_configs.py:
  usage: gap_analysis.py <kernel_trace.csv>   -> per kernel name: count, busy ms; total busy, span, idle inside the steady-state frames."""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt::k_" in r["Kernel_Name"] and "<true>" not in r["Kernel_Name"] and "build_light" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                       # steady state
busy = defaultdict(lambda: [0, 0])
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
    busy[n][0] += 1; busy[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
tot = sum(v[1] for v in busy.values())
for n, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:34s} {c:5d} launches  {t / 1e6:9.3f} ms")
small = sorted(g for g in gaps if g < 200000)
print(f"kernels busy {tot / 1e6:.3f} ms of span {span / 1e6:.3f} ms -> idle {100 * (1 - tot / span):.1f} %;  median gap {small[len(small) // 2] / 1e3:.1f} us, gaps > 20 us: {sum(g > 20000 for g in gaps)} of {len(gaps)}")
