#!/bin/bash
# wave priority (FFH_PRIO build) x HIP stream priorities (FFH_STREAM_PRIOS = main,side,dw) at the per-rank batch, interleaved on one box
R=$(pwd); O=$R/gpurun_out/r4_prio2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
run() {  # label, env assignment, extra bench args...
  local label=$1 envs=$2; shift 2
  L=$(env $envs python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1)
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for sp in "X=1" "FFH_STREAM_PRIOS=-1,0,0" "FFH_STREAM_PRIOS=-1,1,0" "FFH_STREAM_PRIOS=0,1,0" "FFH_STREAM_PRIOS=-1,1,1"; do
    run "prio3 $sp" "$sp" --per-gpu-batch 4096 --steps 100 --warmup 10
    run "prio0 $sp" "$sp" --per-gpu-batch 4096 --steps 100 --warmup 10 "--shim-flags=--backend tools/lab/libffhip_prio0.so"
  done
done
for sp in "X=1" "FFH_STREAM_PRIOS=-1,1,0" "FFH_STREAM_PRIOS=0,1,0"; do
  run "prio3 $sp" "$sp" --steps 20 --warmup 5
  run "prio0 $sp" "$sp" --steps 20 --warmup 5 "--shim-flags=--backend tools/lab/libffhip_prio0.so"
  run "prio3 $sp" "$sp" --workload kaggle --steps 300 --warmup 30
  run "prio3 $sp" "$sp" --workload mlperf --steps 50 --warmup 5
done
FFH_STREAM_PRIOS=-1,1,0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_b4096.txt 2>&1; cat $O/timeline_b4096.txt
find $O/prof -name "*.csv" -size +10M -delete
