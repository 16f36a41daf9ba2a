#!/bin/bash
# kernel trace of the headline model at the per-rank batch of an 8-GPU strong-scaling job (4096 samples, all 26 tables)
R=$(pwd); O=$R/gpurun_out/prof_b4096; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $O/bench.log 2>&1
grep '^{' $O/bench.log | head -c 300; echo
T=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $T > $O/timeline.txt 2>&1
cat $O/timeline.txt
find $O -name "*.csv" -size +10M -delete
