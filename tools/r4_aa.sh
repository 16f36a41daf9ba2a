#!/bin/bash
# round 4 visit AA: old routing / scheduling decisions against their alternatives, whole steps, one box
R=$(pwd); O=$R/gpurun_out/r4_aa; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
L="--backend tools/lab/libffhip_lab.so"
b() { env "$1" python3 bench.py --no-cpu-baseline --no-secondary "${@:3}" "--shim-flags=$2" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
for w in "--steps 20 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10" "--workload mlperf --steps 50 --warmup 5" "--workload kaggle --steps 300 --warmup 30"; do
  echo "lab default            | $w | $(b X=1 "$L" $w)" | tee -a $O/out.txt
  echo "SK_MIN_WEIGHTS_BWD=0   | $w | $(b FFH_SK_MIN_WEIGHTS_BWD=0 "$L" $w)" | tee -a $O/out.txt
  echo "GEMM_NO_GLDS=1         | $w | $(b FFH_GEMM_NO_GLDS=1 "$L" $w)" | tee -a $O/out.txt
  echo "GLDS_NO_DUAL=1         | $w | $(b FFH_GLDS_NO_DUAL=1 "$L" $w)" | tee -a $O/out.txt
  echo "NO_SKINNY=1            | $w | $(b FFH_NO_SKINNY=1 "$L" $w)" | tee -a $O/out.txt
  echo "--no-early-sort        | $w | $(b X=1 "$L --no-early-sort" $w)" | tee -a $O/out.txt
  echo "--no-attach-event      | $w | $(b X=1 "$L --no-attach-event" $w)" | tee -a $O/out.txt
  echo "--serial-dw            | $w | $(b X=1 "$L --serial-dw" $w)" | tee -a $O/out.txt
  echo "--no-overlap           | $w | $(b X=1 "$L --no-overlap" $w)" | tee -a $O/out.txt
done
done
