#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_ae; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for e in "X=1" "FFH_GEMM_CFG=0" "FFH_GEMM_CFG=1" "FFH_GEMM_CFG=2" "FFH_GEMM_NO_GLDS=1"; do echo "== $e" | tee -a $O/out.txt; env $e python3 tools/dw_db_probe.py 8192x512x256 4096x512x256 2>&1 | grep -E "^[0-9]|fwd act relu|dX act relu|dW act relu premasked db|dW act relu db" | tee -a $O/out.txt; done
