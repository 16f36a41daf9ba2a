#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, KiB) per kernel and grid size.
gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane) coalesced read, i.e. HALF the bytes
(MI355X_MICROARCH.md, HBM); the corrected column doubles it."""
import csv, sys, collections, json
out = {}
ALL = "--all" in sys.argv          # every kernel and counter as plain per-launch means (SQ counter passes: tools/pmc_sq.sh)
paths = [a for a in sys.argv[1:] if a != "--all"]
if ALL:
    acc = collections.defaultdict(list)
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, grid, ctr), v in sorted(acc.items()):
        out.setdefault(f"{name} grid={grid}", {})[ctr] = {"launches": len(v), "mean": round(sum(v) / len(v), 1)}
    print(json.dumps(out, indent=1))
    sys.exit(0)
for path in paths:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[(name, r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (name, ctr, grid), v in sorted(acc.items()):
        if not any(k in name for k in ("emb_", "radix")):
            continue
        mean_kib = sum(v) / len(v)
        out.setdefault(f"{name} grid={grid}", {})[ctr] = {"launches": len(v), "mean_KiB": round(mean_kib, 1), "mean_bytes": int(mean_kib * 1024)}
print(json.dumps(out, indent=1))
