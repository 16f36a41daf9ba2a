#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_z; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
python3 tools/gemm_big.py -1,0,1,2 32768x13x512 8192x13x512 32768x256x128 8192x256x128 4096x256x128 2>&1 | grep -v "amdgpu.ids\|kernel library" | tee -a $O/out.txt
for e in "FFH_NO_GLDS=1" ; do echo "== $e" | tee -a $O/out.txt; env $e python3 tools/gemm_big.py -1 32768x256x128 8192x256x128 4096x256x128 8192x512x256 4096x512x256 2>&1 | grep -v "amdgpu.ids\|kernel library" | tee -a $O/out.txt; done
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do for w in "--steps 30 --warmup 5" "--workload mlperf --steps 50 --warmup 5" "--per-gpu-batch 4096 --steps 100 --warmup 10"; do echo "product | $w | $(b $w)" | tee -a $O/out.txt; done; done
