#!/bin/bash
# round 4: smoke, the whole GPU suite, then every judged artefact under profiles/ (tools/refresh_profiles.sh) -- one box visit
R=$(pwd); O=$R/gpurun_out/r4_final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 3300 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_all.log | tail -2 | tee -a $O/summary.txt
bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?" | tee -a $O/summary.txt; tail -25 $O/refresh.log | tee -a $O/summary.txt
