#!/bin/bash
# round 4 visit S: the interaction backward as straight-line code for d = 128 (2 / 4 tiles per pass) vs the general kernel
R=$(pwd); O=$R/gpurun_out/r4_s; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q -k "interaction or dot or mlperf or golden or parity" 2>&1 | grep -E "passed|failed|^FAILED|^E " | tee -a $O/out.txt
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for f in 0 2 4 0 2 4; do
  for B in 8192 32768 2048; do echo "FAST=$f $(FFH_DOT_BWD_FAST=$f python3 tools/dot_bench.py $B 27 128 2>&1 | grep -v 'amdgpu.ids\|kernel library')" | tee -a $O/out.txt; done
done
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1; do
echo "mlperf product (fast 2) $(b --workload mlperf --steps 50 --warmup 5)" | tee -a $O/out.txt
echo "mlperf lab fast 0       $(FFH_DOT_BWD_FAST=0 b --workload mlperf --steps 50 --warmup 5 '--shim-flags=--backend tools/lab/libffhip_lab.so')" | tee -a $O/out.txt
echo "mlperf lab fast 4       $(FFH_DOT_BWD_FAST=4 b --workload mlperf --steps 50 --warmup 5 '--shim-flags=--backend tools/lab/libffhip_lab.so')" | tee -a $O/out.txt
done
