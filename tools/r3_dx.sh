#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "linear or whole_step" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2 3; do python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 %.1f us (graph %s)' % (d['ms_per_step']*1e3, d['config']['step_graph']))"; done
