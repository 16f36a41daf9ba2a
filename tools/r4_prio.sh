#!/bin/bash
# A/B of the wave-priority build (FFH_PRIO=3, the product library) against FFH_PRIO=0 (tools/lab/libffhip_prio0.so), interleaved on one box
R=$(pwd); O=$R/gpurun_out/r4_prio; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2; do
  for v in prio3 prio0; do
    F=""; [ $v = prio0 ] && F="--shim-flags=--backend tools/lab/libffhip_prio0.so"
    for wl in "--per-gpu-batch 4096 --steps 100 --warmup 10" "--steps 20 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange" "--workload kaggle --steps 300 --warmup 30" "--workload mlperf --steps 50 --warmup 5"; do
      L=$(python3 bench.py $wl --no-cpu-baseline --no-secondary "$F" 2>/dev/null | grep '^{' | tail -1)
      echo "$v | $wl | $(echo $L | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $O/summary.txt
    done
  done
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/timeline_b4096_prio3.txt 2>&1; cat $O/timeline_b4096_prio3.txt
find $O/prof -name "*.csv" -size +10M -delete
