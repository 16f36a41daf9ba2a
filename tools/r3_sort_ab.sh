#!/bin/bash
# one box visit: the two-call update's tests, then the step with / without the early sort (fp32 and tensor-op mode), interleaved
out=gpurun_out/s2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round3b.py -x -q > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for r in 1 2; do
  for v in "" "--no-early-sort" "--allow-tensor-op-math-conversion" "--allow-tensor-op-math-conversion --no-early-sort"; do
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 --shim-flags="$v" 2>$out/err.txt | grep "^{" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); u=k.get('embedding_bwd_sgd_fused',{})
print('round $r [%s] %.1f us/step  gather in-step %s  update in-step %s' % ('$v', d['ms_per_step']*1e3, d['roofline'].get('us_per_launch_in_step'), u.get('us_per_launch_in_step')))"
  done
done 2>&1 | tee $out/ab.txt
