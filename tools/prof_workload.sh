#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/prof_workload.sh <workload> [extra bench flags]
WL=$1; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/prof_$WL
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$WL -- python3 bench.py --workload $WL --steps 30 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/prof_$WL/bench.log 2>&1
grep '^{' gpurun_out/prof_$WL/bench.log | head -c 300; echo
S=$(find gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/prof_${WL}_kernel_stats.csv
T=$(find gpurun_out/prof_$WL -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $T > gpurun_out/prof_${WL}_timeline.txt 2>&1
head -50 gpurun_out/prof_${WL}_timeline.txt
find gpurun_out/prof_$WL -name "*.csv" -size +10M -delete
