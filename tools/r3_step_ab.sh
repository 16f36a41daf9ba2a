#!/bin/bash
# A/B of FFConfig flags on the Terabyte step, one box, interleaved: tools/r3_step_ab.sh ROUNDS "flags A" "flags B" ...
rounds=$1; shift
out=gpurun_out/stepab; mkdir -p $out
for r in $(seq $rounds); do
  for v in "$@"; do
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 --shim-flags="$v" 2>$out/err.txt | grep "^{" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print('round $r [%s] %.1f us/step' % ('$v', d['ms_per_step']*1e3))"
  done
done 2>&1 | tee $out/ab.txt
