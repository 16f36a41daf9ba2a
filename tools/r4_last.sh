#!/bin/bash
# round 4, last visit: smoke + the whole GPU suite at the final HEAD
R=$(pwd); O=$R/gpurun_out/r4_last; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tee -a $O/summary.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/summary.txt
