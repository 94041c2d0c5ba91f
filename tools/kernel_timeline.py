#!/usr/bin/env python3
"""kernel_timeline.py kernel_trace.csv [last N]: the last N dispatches of a rocprofv3 --kernel-trace run: name, duration, gap to the one before (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(e - s) / 1e3:8.2f} us  gap {gap:7.2f}  grid {r.get('Grid_Size_X', '?'):>8}  {r['Kernel_Name'][:90]}")
    prev = e
