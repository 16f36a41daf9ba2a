#!/bin/bash
# round 4 visit AC: audit of two more old choices alone: the gather's workgroup cap, the interaction forward's LDS-DMA form
R=$(pwd); O=$R/gpurun_out/r4_ac; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for cap in 1024 512 2048 4096 8192; do echo "FFH_EMB_FWD_CAP=$cap" | tee -a $O/out.txt; FFH_EMB_FWD_CAP=$cap python3 tools/microbench.py emb 2>&1 | grep -E "terabyte|kaggle-26" | cut -c1-100 | tee -a $O/out.txt; done
for v in 0 1; do echo "FFH_DOT_NO_LDS=$v $(FFH_DOT_NO_LDS=$v python3 tools/dot_bench.py 8192 27 128 2>&1 | grep -v 'amdgpu.ids\|kernel library')" | tee -a $O/out.txt; done
