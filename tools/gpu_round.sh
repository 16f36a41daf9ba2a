#!/bin/bash
# one GPU-box visit: A/B of the LDS-DMA GEMM knobs
python -c "import torch" 2>/dev/null
bash tools/ab.sh 3 "FFH_GLDS_SPLIT_ROWS=1" 2>/dev/null | grep round
for v in "FFH_NONE=0" "FFH_GLDS_MAX_TILES=100000" "FFH_NONE=0" "FFH_GLDS_MAX_TILES=100000"; do
  us=$(env $v timeout 300 python bench.py --workload mlperf --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % (d['ms_per_step']*1e3))")
  echo "mlperf $v $us us"
done
for v in "FFH_NONE=0" "FFH_GLDS_MAX_TILES=100000"; do
  us=$(env $v timeout 300 python bench.py --workload terabyte --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % (d['ms_per_step']*1e3))")
  echo "terabyte $v $us us"
done
