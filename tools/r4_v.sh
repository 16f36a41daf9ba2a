#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_v; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for e in "X=1" "FFH_SKINNY_NO_WS=1"; do echo "== $e" | tee -a $O/out.txt; env $e python3 tools/skinny_check.py 2>&1 | grep -v "amdgpu.ids\|kernel library" | tee -a $O/out.txt; done
