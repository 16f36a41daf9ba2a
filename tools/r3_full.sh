#!/bin/bash
O=gpurun_out/r3_full; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
grep -n "passed\|failed\|rc=" $O/pytest_gpu.log | tail -4; tail -2 $O/smoke.log
