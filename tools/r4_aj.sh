#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_aj; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
L="--backend tools/lab/libffhip_lab.so"
b() { env "$1" python3 bench.py --no-cpu-baseline --no-secondary "${@:2}" "--shim-flags=$L" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do for w in "--workload mlperf --steps 50 --warmup 5" "--per-gpu-batch 8192 --steps 60 --warmup 10"; do
echo "dw forked   | $w | $(b X=1 $w)" | tee -a $O/out.txt
echo "dw on s     | $w | $(b FFH_SPLIT_DW_ON_S=1 $w)" | tee -a $O/out.txt
done; done
