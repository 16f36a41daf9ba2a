#!/bin/bash
# round 4, visit D: round-4 GPU tests; deferred big dW (now also for the layer the gradients-ready event rides on); capture of the exchange step
R=$(pwd); O=$R/gpurun_out/r4_d; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests/test_gpu_round4.py -x -q -s > $O/pytest_r4.log 2>&1; echo "pytest round4 rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|fraction beyond" $O/pytest_r4.log | tail -3 | tee -a $O/summary.txt
run() {  # label, env, extra bench args...
  local label=$1 envs=$2; shift 2
  L=$(env $envs python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'), json.dumps(d.get('collectives_in_step_us',{}).get('rank0')))")" | tee -a $O/summary.txt
}
LAB="--backend tools/lab/libffhip_lab.so"
for rep in 1 2 3; do
  for b in 4096 8192; do
    run "auto      " X=1 --per-gpu-batch $b --steps 100 --warmup 10
    run "defer off " X=1 --per-gpu-batch $b --steps 100 --warmup 10 --shim-flags=--no-defer-big-dw
  done
  run "lab default" X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 "--shim-flags=$LAB"
  run "lab nosplit" FFH_SK_NO_SPLIT=1 --per-gpu-batch 4096 --steps 100 --warmup 10 "--shim-flags=$LAB"
  run "exch auto " X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange
  run "exch nodef" X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange --shim-flags=--no-defer-big-dw
  run "exch graph" X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange --force-graph --shim-flags=--capture-exchange
done
run "32768 auto" X=1 --steps 20 --warmup 5
run "32768 defer" X=1 --steps 20 --warmup 5 --shim-flags=--defer-big-dw
run "mlperf     " X=1 --workload mlperf --steps 50 --warmup 5
run "mlperf nodef" X=1 --workload mlperf --steps 50 --warmup 5 --shim-flags=--no-defer-big-dw
run "kaggle exch graph" X=1 --workload kaggle --steps 300 --warmup 30 --force-exchange --force-graph --shim-flags=--capture-exchange
run "kaggle exch eager" X=1 --workload kaggle --steps 300 --warmup 30 --force-exchange
for v in plain exch; do
  F="--per-gpu-batch 4096"; [ $v = exch ] && F="--per-gpu-batch 4096 --force-exchange"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -- python3 bench.py $F --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_$v.log 2>&1
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_$v.txt 2>&1
  find $O/prof_$v -name "*.csv" -size +10M -delete
done
cat $O/timeline_plain.txt
