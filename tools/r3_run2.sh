#!/bin/bash
O=gpurun_out/r3_run2; mkdir -p $O
L=tools/lab/gemm_sk_lab
{
timeout 120 $L 32768 1024 1024 0 256 1
timeout 120 $L 32768 1024 1024 1 256 1
timeout 120 $L 32768 3456 1024 1 256 0
timeout 120 $L 32768 1024 512 1 256 0
cd /tmp; FFH_GEMM_CFG=-1 timeout 300 python3 $GRAFT_REPO_ROOT/tools/gemm_big.py child 32768x1024x1024 32768x3456x1024 2>&1 | grep -v "DLRM\|amdgpu.ids"
GEMM_BIG_RELU_X=1 FFH_GEMM_CFG=-1 timeout 300 python3 $GRAFT_REPO_ROOT/tools/gemm_big.py child 32768x1024x1024 32768x3456x1024 2>&1 | grep -v "DLRM\|amdgpu.ids"
} > $O/sk_lab2.txt 2>&1
cat $O/sk_lab2.txt
