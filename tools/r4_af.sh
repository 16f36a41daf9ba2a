#!/bin/bash
# round 4 visit AF: data gradient on the persistent kernel also where the layer's weight gradient is not one of its shapes
R=$(pwd); O=$R/gpurun_out/r4_af; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tee -a $O/out.txt
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do for w in "--workload mlperf --steps 50 --warmup 5" "--steps 20 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10" "--per-gpu-batch 8192 --steps 60 --warmup 10" "--workload kaggle --steps 300 --warmup 30"; do echo "product | $w | $(b $w)" | tee -a $O/out.txt; done; done
