#!/bin/bash
# A/B of shim flags on the exchange path (1-rank RCCL group), interleaved: tools/ab_exchange.sh ROUNDS "--no-dx-scatter" ...
rounds=$1; shift
variants=("" "$@")
for r in $(seq $rounds); do
  for v in "${variants[@]}"; do
    us=$(timeout 200 python bench.py --force-exchange --no-cpu-baseline --steps 400 --shim-flags="$v" 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us' % (d['ms_per_step']*1e3))")
    echo "round $r  [${v:-default}]  $us"
  done
done
