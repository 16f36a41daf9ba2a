#!/bin/bash
# round 4 visit O: layers with few k-tiles per workgroup on the persistent kernels vs the LDS-DMA / register-staged kernels (lab switches)
R=$(pwd); O=$R/gpurun_out/r4_o; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
S="8192x512x256 4096x512x256 4096x1024x512 8192x256x128 16384x512x256"
for e in "X=1" "FFH_SK_NO_SPLIT=1" "FFH_GEMM_NO_SK=1"; do
  echo "== $e" | tee -a $O/out.txt
  env $e python3 tools/gemm_big.py -1 $S 2>&1 | grep -v "amdgpu.ids\|FFH_GEMM_CFG" | tee -a $O/out.txt
done
