#!/bin/bash
# round 4 visit U: narrow-layer backward with partial rows + last arriver instead of atomic chains
R=$(pwd); O=$R/gpurun_out/r4_u; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "linear or Linear or skinny or fuzz or golden or step or kaggle or pair" 2>&1 | grep -E "passed|failed|^FAILED|^E " | tee -a $O/out.txt
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for e in "FFH_SKINNY_NO_WS=1" "X=1" "FFH_SKINNY_NBLK=128" "FFH_SKINNY_NBLK=256" "FFH_SKINNY_NBLK=512"; do
  echo "$e" | tee -a $O/out.txt
  env $e python3 tools/dw_db_probe.py 32768x256x1 4096x256x1 2048x256x1 2048x64x16 2>&1 | grep -E "^[0-9]|dW act none db|dW act relu db|dX act none|check" | tee -a $O/out.txt
done
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for w in "--steps 30 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10" "--workload mlperf --steps 50 --warmup 5" "--workload kaggle --steps 300 --warmup 30"; do
echo "product | $w | $(b $w)" | tee -a $O/out.txt
echo "no ws   | $w | $(FFH_SKINNY_NO_WS=1 b $w '--shim-flags=--backend tools/lab/libffhip_lab.so')" | tee -a $O/out.txt
done
done
