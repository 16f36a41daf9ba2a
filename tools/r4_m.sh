#!/bin/bash
# round 4 visit M: wave priority 3 by default + the embedding stream at a higher HIP priority (ABI 11): whole GPU suite, then every workload with / without
R=$(pwd); O=$R/gpurun_out/r4_m; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_all.log | tail -2 | tee -a $O/summary.txt
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for w in "--steps 30 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10" "--workload mlperf --steps 50 --warmup 5" "--workload kaggle --steps 300 --warmup 30" "--workload giant --steps 200 --warmup 20" "--per-gpu-batch 4096 --steps 100 --warmup 10 --force-exchange" "--workload kaggle --steps 300 --warmup 30 --force-exchange"; do
echo "default     | $w | $(b $w)" | tee -a $O/summary.txt
echo "no strm prio | $w | $(b $w --shim-flags=--no-stream-priorities)" | tee -a $O/summary.txt
done
done
