#!/bin/bash
# round 4 visit L: wave priority 3 in the non-persistent kernels again, now that the weight-gradient GEMMs run without the bias sums
R=$(pwd); O=$R/gpurun_out/r4_l; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
b() { env "$1" python3 bench.py --no-cpu-baseline --no-secondary "${@:2}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for sp in "X=1" "FFH_STREAM_PRIOS=0,-1,0" "FFH_STREAM_PRIOS=0,-1,1"; do
echo "32768 prio0 $sp $(b $sp --steps 30 --warmup 5)" | tee -a $O/summary.txt
echo "32768 prio3 $sp $(b $sp --steps 30 --warmup 5 '--shim-flags=--backend tools/lab/libffhip_prio3.so')" | tee -a $O/summary.txt
done
echo "mlperf prio0 $(b X=1 --workload mlperf --steps 50 --warmup 5)" | tee -a $O/summary.txt
echo "mlperf prio3 $(b X=1 --workload mlperf --steps 50 --warmup 5 '--shim-flags=--backend tools/lab/libffhip_prio3.so')" | tee -a $O/summary.txt
echo "4096 prio0 $(b X=1 --per-gpu-batch 4096 --steps 100 --warmup 10)" | tee -a $O/summary.txt
echo "4096 prio3 $(b X=1 --per-gpu-batch 4096 --steps 100 --warmup 10 '--shim-flags=--backend tools/lab/libffhip_prio3.so')" | tee -a $O/summary.txt
done
