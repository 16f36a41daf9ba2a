#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel stats and the timeline of one step."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step is delimited by the kernel that opens backward(): the loss + metrics launch, or the last layer's one-launch
# backward that now carries the loss step (linear_skinny_bwd_kernel<*, 1> / <*, 4> right after a skinny forward)
idx = [i for i, r in enumerate(rows) if "metrics_kernel" in r["Kernel_Name"]]
if len(idx) < 25:
    idx = [i for i, r in enumerate(rows) if i > 0 and "linear_skinny_bwd_kernel" in r["Kernel_Name"] and ("linear_skinny_fwd_kernel" in rows[i - 1]["Kernel_Name"] or "mlp_chain_fwd_kernel" in rows[i - 1]["Kernel_Name"])]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -20
a, b = idx[which], idx[which + 1]
# a step spans from the first kernel after previous step's last kernel ... approximate: window between consecutive mse kernels
t0 = int(rows[a]["Start_Timestamp"])
print(f"window between two consecutive loss (+ last layer backward) kernels: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
busy = 0
last_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    gx = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - last_end) / 1e3:6.1f}  grid {gx}/{wg}  {name}")
    busy += e - s
    last_end = max(last_end, e)
print(f"sum of kernel durations in window: {busy / 1e3:.1f} us")
