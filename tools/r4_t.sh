#!/bin/bash
# round 4 visit T: new d = 128 interaction parity cases; the 256 -> 1 backward vs the number of workgroups (atomic chains per weight)
R=$(pwd); O=$R/gpurun_out/r4_t; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dot_interaction" 2>&1 | grep -E "passed|failed|^FAILED|^E " | tee -a $O/out.txt
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for n in 0 32 64 128 256 512 1024; do
  echo "FFH_SKINNY_NBLK=$n" | tee -a $O/out.txt
  FFH_SKINNY_NBLK=$n python3 tools/dw_db_probe.py 32768x256x1 4096x256x1 2>&1 | grep -E "^[0-9]|dW act none db|dX act none|fwd act none" | tee -a $O/out.txt
done
