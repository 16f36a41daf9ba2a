"""Prints VGPR / SGPR / scratch / LDS / occupancy of every kernel (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "dlrm_flexflow_amd", "csrc")
for f in sorted(os.listdir(CSRC)):
    if not f.endswith(".hip"):
        continue
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                        "-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, f), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()[:90]}
        cur[k] = v
        if k.startswith("LDS Size"):
            print(f"{f:16s} {cur['name']:90s} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} sgpr={cur.get('TotalSGPRs')} "
                  f"scratch={cur.get('ScratchSize [bytes/lane]')} occ={cur.get('Occupancy [waves/SIMD]')} lds={v}")
