#!/bin/bash
# A/B of shim flags on ONE box, interleaved rounds: tools/ab_flags.sh ROUNDS "--serial-dw" "--no-overlap" ...
rounds=$1; shift
variants=("" "$@")
for r in $(seq $rounds); do
  for v in "${variants[@]}"; do
    us=$(timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 400 --shim-flags="$v" 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us (graph %s)' % (d['ms_per_step']*1e3, d['config']['step_graph']))")
    echo "round $r  [${v:-default}]  $us"
  done
done
