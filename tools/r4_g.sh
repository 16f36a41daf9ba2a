#!/bin/bash
# round 4, visit G: order of the biggest layer's two GEMMs (A/B), capture-exchange with the collectives on the capturing stream
R=$(pwd); O=$R/gpurun_out/r4_g; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
run() {  # label, extra bench args...
  local label=$1; shift
  L=$(python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'))")" | tee -a $O/summary.txt
}
for rep in 1 2 3; do
  for b in 4096 8192; do
    run "mode 0" --per-gpu-batch $b --steps 100 --warmup 10
    run "mode 1" --per-gpu-batch $b --steps 100 --warmup 10 "--shim-flags=--big-dw-mode 1"
    run "mode 2" --per-gpu-batch $b --steps 100 --warmup 10 "--shim-flags=--big-dw-mode 2"
  done
  run "mode 0" --steps 20 --warmup 5
  run "mode 1" --steps 20 --warmup 5 "--shim-flags=--big-dw-mode 1"
  run "mode 2" --steps 20 --warmup 5 "--shim-flags=--big-dw-mode 2"
  run "mlperf mode 0" --workload mlperf --steps 50 --warmup 5
  run "mlperf mode 1" --workload mlperf --steps 50 --warmup 5 "--shim-flags=--big-dw-mode 1"
  run "mlperf mode 2" --workload mlperf --steps 50 --warmup 5 "--shim-flags=--big-dw-mode 2"
done
run "kaggle exch graph no-overlap" --workload kaggle --steps 300 --warmup 30 --force-exchange --force-graph "--shim-flags=--capture-exchange --no-overlap"
run "kaggle exch eager no-overlap" --workload kaggle --steps 300 --warmup 30 --force-exchange "--shim-flags=--no-overlap"
for m in 1 2; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_m$m -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary "--shim-flags=--big-dw-mode $m" > $O/bench_m$m.log 2>&1
  T=$(find $O/prof_m$m -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_b4096_mode$m.txt 2>&1
  find $O/prof_m$m -name "*.csv" -size +10M -delete
done
cat $O/timeline_b4096_mode1.txt
