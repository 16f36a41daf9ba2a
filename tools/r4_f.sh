#!/bin/bash
# round 4, visit F: two weight-gradient streams (A/B), capture-exchange crash under rocgdb, sort microbenchmark back on the round-3 kernels, whole GPU suite
R=$(pwd); O=$R/gpurun_out/r4_f; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 tools/microbench.py emb terabyte-4tables terabyte-26 2>&1 | grep -v amdgpu.ids | tee $O/emb.txt
run() {  # label, extra bench args...
  local label=$1; shift
  L=$(python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>$O/last.err | grep '^{' | tail -1)
  if [ -z "$L" ]; then echo "$label | $* | FAILED: $(tail -3 $O/last.err | tr '\n' ' ')" | tee -a $O/summary.txt; return; fi
  echo "$label | $* | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'))")" | tee -a $O/summary.txt
}
for rep in 1 2 3; do
  for b in 4096 8192; do
    run "two streams" --per-gpu-batch $b --steps 100 --warmup 10
    run "one stream " --per-gpu-batch $b --steps 100 --warmup 10 --shim-flags=--one-dw-stream
  done
  run "two streams" --steps 20 --warmup 5
  run "one stream " --steps 20 --warmup 5 --shim-flags=--one-dw-stream
  run "two streams" --workload mlperf --steps 50 --warmup 5
  run "one stream " --workload mlperf --steps 50 --warmup 5 --shim-flags=--one-dw-stream
  run "two streams" --workload kaggle --steps 300 --warmup 30
  run "one stream " --workload kaggle --steps 300 --warmup 30 --shim-flags=--one-dw-stream
  run "two streams exch" --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange
  run "one stream  exch" --per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange --shim-flags=--one-dw-stream
done
for v in b4096 b32768; do
  F="--per-gpu-batch 4096 --steps 30 --warmup 5"; [ $v = b32768 ] && F="--steps 8 --warmup 3"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -- python3 bench.py $F --no-cpu-baseline --no-secondary > $O/bench_$v.log 2>&1
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T -3 > $O/timeline_$v.txt 2>&1
  find $O/prof_$v -name "*.csv" -size +10M -delete
done
cat $O/timeline_b4096.txt
NCCL_DEBUG=WARN timeout 600 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info threads" --args python3 bench.py --workload kaggle --steps 20 --warmup 5 --force-exchange --force-graph "--shim-flags=--capture-exchange" --no-cpu-baseline --no-secondary > $O/capture_gdb.log 2>&1
grep -n "SIGSEGV\|^#[0-9]" $O/capture_gdb.log | head -40 | tee -a $O/summary.txt
timeout 3000 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round4.py::test_exchange_step_captured_as_a_graph_equals_eager_bit_for_bit > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|FATAL" $O/pytest_all.log | tail -3 | tee -a $O/summary.txt
