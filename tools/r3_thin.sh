#!/bin/bash
out=gpurun_out/thin; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py -x -q -k "linear or thin or first_layer or whole_step" > $out/tests.txt 2>&1; grep -E "passed|failed|rror" $out/tests.txt | tail -3
for r in 1 2; do for v in "FFH_THIN_ROWS_MIN_BATCH=8192" ; do
python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | grep "^{" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 %.1f us/step' % (d['ms_per_step']*1e3))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
grep -E "thin|skinny|sgd_kernel|act_bwd" $GRAFT_REPO_ROOT/$out/prof/t_kernel_stats.csv | cut -c1-60,100-200
