#!/bin/bash
# round 4 visit AB: where the index-only sort of the table update goes at big per-GPU batches: in front of the apply phase (default), behind the gather, at the start of backward
R=$(pwd); O=$R/gpurun_out/r4_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
for w in "--steps 20 --warmup 5" "--workload mlperf --steps 50 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10" "--per-gpu-batch 8192 --steps 60 --warmup 10"; do
  echo "by shape                | $w | $(b $w)" | tee -a $O/out.txt
  echo "--early-sort            | $w | $(b $w --shim-flags=--early-sort)" | tee -a $O/out.txt
  echo "--no-early-sort         | $w | $(b $w --shim-flags=--no-early-sort)" | tee -a $O/out.txt
  echo "--sort-at-backward-start | $w | $(b $w --shim-flags=--sort-at-backward-start)" | tee -a $O/out.txt
done
done
