#!/bin/bash
# PMC passes over the stand-alone split-GEMM lab (tools/lab/gemm_x3_lab): SQ issue / wait split, LDS, L2 hit rate, memory-side reads.
# usage: tools/pmc_x3_lab.sh [batch in out]   (counter-only passes: no trace domains beside --pmc)
R=$(pwd); O=$R/gpurun_out/pmc_x3_lab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
B=${1:-32768}; I=${2:-3456}; U=${3:-1024}
run() { timeout 300 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- ./tools/lab/gemm_x3_lab $B $I $U 0 0 3 > $O/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
run c "FETCH_SIZE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
run d "WRITE_SIZE TCC_REQ_sum SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
python3 tools/pmc_summary.py --all $(find $O -name "*counter_collection.csv") > $O/summary.json
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "gemm" in k: print(k[:60], {c: round(x["mean"]) for c,x in v.items()})
PY
