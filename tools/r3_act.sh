#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_bf16_mode.py tests/test_gpu_fuzz.py -x -q -k "linear or bf16 or mode" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2; do
python bench.py --no-cpu-baseline --no-secondary --steps 200 --shim-flags=--allow-tensor-op-math-conversion 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 %.1f us' % (d['ms_per_step']*1e3))"
python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 %.1f us' % (d['ms_per_step']*1e3))"
done
