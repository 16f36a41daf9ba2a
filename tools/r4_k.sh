#!/bin/bash
# round 4 visit K: the lower layer's bias gradient from the upper layer's dX epilogue (ABI 10): parity subset, benches with / without
R=$(pwd); O=$R/gpurun_out/r4_k; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "linear or Linear or gemm or route or per_rank or mlperf or three_steps or bias_gradient or golden or step" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_sub.log | tee -a $O/summary.txt
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
echo "32768 colsum    $(b --steps 30 --warmup 5)" | tee -a $O/summary.txt
echo "32768 no colsum $(b --steps 30 --warmup 5 --shim-flags=--no-dx-colsum)" | tee -a $O/summary.txt
echo "4096 colsum     $(b --per-gpu-batch 4096 --steps 100 --warmup 10)" | tee -a $O/summary.txt
echo "4096 no colsum  $(b --per-gpu-batch 4096 --steps 100 --warmup 10 --shim-flags=--no-dx-colsum)" | tee -a $O/summary.txt
echo "mlperf colsum   $(b --workload mlperf --steps 50 --warmup 5)" | tee -a $O/summary.txt
echo "mlperf no colsum $(b --workload mlperf --steps 50 --warmup 5 --shim-flags=--no-dx-colsum)" | tee -a $O/summary.txt
echo "kaggle colsum   $(b --workload kaggle --steps 300 --warmup 30)" | tee -a $O/summary.txt
echo "kaggle no colsum $(b --workload kaggle --steps 300 --warmup 30 --shim-flags=--no-dx-colsum)" | tee -a $O/summary.txt
done
