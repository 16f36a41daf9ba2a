#!/bin/bash
# SQ counters of the split-bf16x3 forward GEMM alone (counter-only passes).  FFH_X3_WS in the environment picks the kernel.
R=$(pwd); O=$R/gpurun_out/pmc_x3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
M=${1:-2}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 tools/x3_probe.py $M > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 tools/x3_probe.py $M > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -- python3 tools/x3_probe.py $M > $O/c.log 2>&1
for f in $O/a.log $O/b.log $O/c.log; do tail -n 1 $f; done
python3 tools/pmc_summary.py --all $(find $O -name "*counter_collection.csv") > $O/summary.json
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "gemm" in k: print(k[:70], {c: round(x["mean"]) for c,x in v.items()})
PY
