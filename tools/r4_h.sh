#!/bin/bash
# round 4, visit H: CUs left free by the biggest layer's persistent weight-gradient GEMM (A/B)
R=$(pwd); O=$R/gpurun_out/r4_h; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
run() {  # label, extra bench args...
  local label=$1; shift
  L=$(python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'))")" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for n in 0 16 32 48 64; do
    run "reserve $n" --per-gpu-batch 4096 --steps 100 --warmup 10 "--shim-flags=--dw-cu-reserve $n"
  done
  for n in 0 16 32; do
    run "reserve $n" --per-gpu-batch 8192 --steps 100 --warmup 10 "--shim-flags=--dw-cu-reserve $n"
    run "reserve $n" --steps 20 --warmup 5 "--shim-flags=--dw-cu-reserve $n"
    run "mlperf reserve $n" --workload mlperf --steps 50 --warmup 5 "--shim-flags=--dw-cu-reserve $n"
    run "exch reserve $n" --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange "--shim-flags=--dw-cu-reserve $n"
  done
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary "--shim-flags=--dw-cu-reserve 32" > $O/bench_prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_b4096_reserve32.txt 2>&1; cat $O/timeline_b4096_reserve32.txt
find $O/prof -name "*.csv" -size +10M -delete
