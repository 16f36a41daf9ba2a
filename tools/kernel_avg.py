#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: average duration of one kernel over all launches, and over its longest run of
back-to-back launches (bench.py's per-kernel probe: the interval its `roofline.us_per_launch` is measured on)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
best, cur = [], []
for r in rows:
    if pat in r["Kernel_Name"]:
        cur.append(r)
    else:
        if len(cur) > len(best): best = cur
        cur = []
if len(cur) > len(best): best = cur
allr = [r for r in rows if pat in r["Kernel_Name"]]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{pat}: {len(allr)} launches, average {sum(map(dur, allr)) / len(allr) / 1e3:.2f} us")
if best:
    span = (int(best[-1]["End_Timestamp"]) - int(best[0]["Start_Timestamp"])) / len(best)
    print(f"  longest back-to-back run: {len(best)} launches, average kernel duration {sum(map(dur, best)) / len(best) / 1e3:.2f} us, "
          f"start-to-start {span / 1e3:.2f} us per launch (under the profiler)")
