#!/bin/bash
# round 4 visit N: the stream-K weight gradient with the fix-up of the SPLIT form instead of an atomic epilogue per segment (lab switch)
R=$(pwd); O=$R/gpurun_out/r4_n; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so DW_CHECK=1
for mx in 0 64 200; do
  echo "FFH_SK_DW_SPLIT_MAX_IT=$mx" | tee -a $O/dw.txt
  FFH_SK_DW_SPLIT_MAX_IT=$mx timeout 600 python3 tools/dw_db_probe.py 4096x3456x1024 4096x1024x1024 4096x1024x512 8192x1024x1024 8192x512x1024 8192x1024x512 2>&1 | grep -v amdgpu.ids | grep -E "^[0-9]|no db|check" | tee -a $O/dw.txt
done
