#!/bin/bash
# SQ counters of the step's kernels: two rocprofv3 --pmc passes (counters only, no trace domains) over a short bench run.
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_sq
cd $R
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-trace > gpurun_out/pmc_sq/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-trace > gpurun_out/pmc_sq/b.log 2>&1
tail -2 gpurun_out/pmc_sq/a.log gpurun_out/pmc_sq/b.log
python3 tools/pmc_summary.py --all $(find gpurun_out/pmc_sq -name "*counter_collection.csv") > gpurun_out/pmc_sq/summary.json
find gpurun_out/pmc_sq -name "*.csv" -size +20M -delete
head -c 3000 gpurun_out/pmc_sq/summary.json
