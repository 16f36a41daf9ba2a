#!/bin/bash
# round 4 visit J: bias gradient taken in turns inside the stream-K dW (parity subset, the probe, the benches)
R=$(pwd); O=$R/gpurun_out/r4_j; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "linear or Linear or gemm or route or per_rank or mlperf or three_steps" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_sub.log | tee -a $O/summary.txt
python3 tools/dw_db_probe.py 4096x3456x1024 4096x1024x1024 4096x1024x512 32768x3456x1024 32768x1024x1024 32768x1024x512 8192x1024x1024 8192x512x1024 2>&1 | grep -v amdgpu.ids | grep -E "^[0-9]|dW act (none|relu premasked)" > $O/dw_db.txt; cat $O/dw_db.txt
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32768', d['value'], d['ms_per_step'])" | tee -a $O/summary.txt
python3 bench.py --no-cpu-baseline --no-secondary --per-gpu-batch 4096 --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4096', d['value'], d['ms_per_step'])" | tee -a $O/summary.txt
python3 bench.py --no-cpu-baseline --no-secondary --workload mlperf --steps 50 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlperf', d['value'], d['ms_per_step'])" | tee -a $O/summary.txt
done
