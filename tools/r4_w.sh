#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_w; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 2400 python3 -m pytest tests -m gpu -q -k "ranks_on_one_gpu or linear or Linear or skinny or fuzz or golden or step or kaggle or pair" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tee -a $O/out.txt
