#!/bin/bash
# round 4, visit B: the new GPU tests, then the whole GPU suite; A/B of the update-behind-bottom-backward schedule; timelines
R=$(pwd); O=$R/gpurun_out/r4_b; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests/test_gpu_round4.py -x -q > $O/pytest_r4.log 2>&1; echo "pytest round4 rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_r4.log | tee -a $O/summary.txt
run() {  # label, extra bench args...
  local label=$1; shift
  L=$(python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d.get('collectives_in_step_us',{}).get('rank0')))")" | tee -a $O/summary.txt
}
for rep in 1 2; do
  run "auto" --per-gpu-batch 4096 --steps 100 --warmup 10
  run "off " --per-gpu-batch 4096 --steps 100 --warmup 10 --shim-flags=--no-update-behind-bottom-bwd
  run "auto" --steps 20 --warmup 5
  run "off " --steps 20 --warmup 5 --shim-flags=--no-update-behind-bottom-bwd
  run "auto" --workload mlperf --steps 50 --warmup 5
  run "off " --workload mlperf --steps 50 --warmup 5 --shim-flags=--no-update-behind-bottom-bwd
  run "auto" --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange
  run "off " --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange --shim-flags=--no-update-behind-bottom-bwd
  run "auto" --workload kaggle --steps 300 --warmup 30
  run "on  " --workload kaggle --steps 300 --warmup 30 --shim-flags=--update-behind-bottom-bwd
done
for v in plain exch; do
  F=""; [ $v = exch ] && F="--force-exchange"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary $F > $O/bench_$v.log 2>&1
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_b4096_$v.txt 2>&1
  find $O/prof_$v -name "*.csv" -size +10M -delete
done
cat $O/timeline_b4096_plain.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round4.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_all.log | tee -a $O/summary.txt
