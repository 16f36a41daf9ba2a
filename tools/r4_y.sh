#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_y; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for e in "X=1" "FFH_THIN_ROWS_MIN_BATCH=100000000" "FFH_NO_THIN=1"; do echo "== $e" | tee -a $O/out.txt; env $e python3 tools/dw_db_probe.py 32768x13x512 8192x13x512 4096x13x512 2>&1 | grep -E "^[0-9]|fwd act|dW act relu premasked db" | tee -a $O/out.txt; done
