#!/bin/bash
# SQ counters of the big fp32 GEMMs alone (this repo's kernels and hipBLASLt's in one process), one layer shape per pass pair:
# matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), LDS bank conflicts,
# wait buckets.  Counter-only rocprofv3 passes (no trace domains); the program directly after "--".
R=$(pwd); O=$R/gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
for SH in 32768x1024x1024 32768x3456x1024; do
  FFH_GEMM_CFG=-1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a_$SH -- python3 tools/gemm_big.py child $SH > $O/a_$SH.log 2>&1
  FFH_GEMM_CFG=-1 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b_$SH -- python3 tools/gemm_big.py child $SH > $O/b_$SH.log 2>&1
  python3 tools/pmc_summary.py --all $(find $O/a_$SH $O/b_$SH -name "*counter_collection.csv") > $O/raw_$SH.json
done
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "pmc_gemm")
out = {"method": "two counter-only rocprofv3 --pmc passes over `python3 tools/gemm_big.py child <shape>` per layer shape (tools/pmc_gemm.sh); "
       "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
       "kernels whose name starts with Cijk_ are hipBLASLt's (torch.mm on the same operands in the same process)"}
for f in sorted(glob.glob(os.path.join(O, "raw_*.json"))):
    shape = os.path.basename(f)[4:-5]
    d = json.load(open(f))
    blk = {}
    for k, v in d.items():
        if not any(t in k for t in ("gemm_", "Cijk_")):
            continue
        g = v.get("GRBM_GUI_ACTIVE", {}).get("mean"); m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean")
        e = {c: x["mean"] for c, x in v.items()}
        if g and m:
            e["mfma_busy"] = round(m / (g / 8.0 * 1024.0), 3)
        if v.get("SQ_LDS_IDX_ACTIVE", {}).get("mean"):
            e["lds_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", {}).get("mean", 0.0) / v["SQ_LDS_IDX_ACTIVE"]["mean"], 4)
        blk[k[:110]] = e
    out[shape] = blk
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for shape, blk in out.items():
    if shape == "method": continue
    for k, e in blk.items():
        print(shape, k[:70], "mfma_busy", e.get("mfma_busy"), "lds_conflict_share", e.get("lds_conflict_share"))
PY
find $O -name "*.csv" -size +5M -delete
