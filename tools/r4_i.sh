#!/bin/bash
# round 4 visit I: what the bias gradient costs the stream-K dW; the bottom MLP's small layers alone
R=$(pwd); O=$R/gpurun_out/r4_i; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 tools/dw_db_probe.py 4096x3456x1024 4096x1024x1024 4096x1024x512 4096x512x256 32768x3456x1024 32768x1024x1024 8192x1024x1024 2>&1 | grep -v amdgpu.ids > $O/dw_db.txt
python3 tools/dw_db_probe.py 4096x256x128 4096x13x512 4096x256x1 32768x512x256 32768x256x128 32768x13x512 2>&1 | grep -v amdgpu.ids > $O/small.txt
cat $O/dw_db.txt $O/small.txt
