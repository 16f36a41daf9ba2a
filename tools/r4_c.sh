#!/bin/bash
# round 4, visit C: stream-K split GEMMs, deferred big dW, padded 479 layer: tests, microbenchmarks, A/B, timelines
R=$(pwd); O=$R/gpurun_out/r4_c; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests/test_gpu_round4.py -x -q > $O/pytest_r4.log 2>&1; echo "pytest round4 rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_r4.log | tail -2 | tee -a $O/summary.txt
python3 tools/gemm_big.py -1 4096x3456x1024 4096x1024x1024 4096x1024x512 8192x512x1024 8192x1024x1024 8192x1024x512 8192x512x256 2>&1 | grep -v "DLRM\|amdgpu.ids" | tee $O/gemm_split.txt
FFH_SK_NO_SPLIT=1 python3 tools/gemm_big.py -1 4096x3456x1024 4096x1024x512 8192x512x256 2>&1 | grep -v "DLRM\|amdgpu.ids" | tee $O/gemm_nosplit.txt
run() {  # label, env, extra bench args...
  local label=$1 envs=$2; shift 2
  L=$(env $envs python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $O/summary.txt
}
for rep in 1 2; do
  for b in 4096 8192 16384; do
    run "auto          " X=1 --per-gpu-batch $b --steps 100 --warmup 10
    run "defer off     " X=1 --per-gpu-batch $b --steps 100 --warmup 10 --shim-flags=--no-defer-big-dw
    run "defer on      " X=1 --per-gpu-batch $b --steps 100 --warmup 10 --shim-flags=--defer-big-dw
    run "nosplit auto  " FFH_SK_NO_SPLIT=1 --per-gpu-batch $b --steps 100 --warmup 10
  done
  run "auto          " X=1 --steps 20 --warmup 5
  run "defer on      " X=1 --steps 20 --warmup 5 --shim-flags=--defer-big-dw
  run "mlperf        " X=1 --workload mlperf --steps 50 --warmup 5
  run "mlperf nopad  " X=1 --workload mlperf --steps 50 --warmup 5 --shim-flags=--no-pad-linear-k
  run "mlperf defer  " X=1 --workload mlperf --steps 50 --warmup 5 --shim-flags=--defer-big-dw
  run "exch auto     " X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange
  run "exch deferoff " X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange --shim-flags=--no-defer-big-dw
done
for v in plain exch mlperf; do
  F="--per-gpu-batch 4096"; [ $v = exch ] && F="--per-gpu-batch 4096 --force-exchange"; [ $v = mlperf ] && F="--workload mlperf"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -- python3 bench.py $F --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_$v.log 2>&1
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_$v.txt 2>&1
  find $O/prof_$v -name "*.csv" -size +10M -delete
done
cat $O/timeline_plain.txt
