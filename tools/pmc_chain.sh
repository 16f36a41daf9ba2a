#!/bin/bash
# SQ counters of the chain kernels alone (counter-only rocprofv3 passes over tools/chain_bench.py): bash tools/pmc_chain.sh <batch> [chain_bench args]
R=$(pwd); O=$R/gpurun_out/pmc_chain; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
B=${1:-4096}; shift
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 tools/chain_bench.py $B "$@" > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- python3 tools/chain_bench.py $B "$@" > $O/b.log 2>&1
python3 tools/pmc_summary.py --all $(find $O -name "*counter_collection.csv") > $O/summary.json
find $O -name "*.csv" -size +20M -delete
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "chain" in k: print(k, {c:x["mean"] for c,x in v.items()})
PY
