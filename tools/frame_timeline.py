#!/usr/bin/env python3
"""Timeline of the kernels of ONE steady-state frame from a rocprofv3 --kernel-trace CSV: start offset, duration, idle gap before each.
  usage: frame_timeline.py <kernel_trace.csv> <name of the frame's first kernel, e.g. k_gi_primary>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "<true>" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0, prev = int(rows[a]["Start_Timestamp"]), None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]}  grid {r.get('Grid_Size', '?')}")
    prev = e
print(f"frame span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
