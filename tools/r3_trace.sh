#!/bin/bash
# one steady-state step of the default bench under the kernel trace -> gpurun_out/r3_trace/step_timeline.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r3_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-trace $BENCH_EXTRA 2>/dev/null | grep '^{' | tail -1 > $O/bench.json
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T -3 > $O/step_timeline.txt
find $O/prof -name "*.csv" -size +5M -delete
cat $O/step_timeline.txt
