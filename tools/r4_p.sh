#!/bin/bash
# round 4 visit P: the whole GPU suite again (two test shapes moved to >= 8 k-tiles per workgroup)
R=$(pwd); O=$R/gpurun_out/r4_p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_all.log | tail -2 | tee -a $O/summary.txt; grep -E "^FAILED" $O/pytest_all.log | tee -a $O/summary.txt
