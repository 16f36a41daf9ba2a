#!/bin/bash
# Kernel timeline of one step of `bench.py --shim-flags=$2 [more bench flags]` -> gpurun_out/$1_step_timeline.txt (the step's line on stdout first)
# usage (on the GPU box): tools/step_timeline.sh r06_split --fp32-split-bf16x3 [--per-gpu-batch 4096 ...]
N=$1; F=$2; shift 2
R=$(pwd); O=$R/gpurun_out/prof_$N; rm -rf $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --shim-flags=$F "$@" 2>/dev/null | grep "^{" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$N', d['value'], d['ms_per_step'])"
T=$(find $O -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > gpurun_out/${N}_step_timeline.txt; rm -rf $O
