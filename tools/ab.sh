#!/bin/bash
# A/B of env-var toggles on ONE box, interleaved rounds, best of each: tools/ab.sh ROUNDS "VAR1=1" "VAR2=1 VAR3=0" ...
rounds=$1; shift
declare -A best
variants=("FFH_NONE=0" "$@")
for r in $(seq $rounds); do
  for v in "${variants[@]}"; do
    us=$(env $v timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 400 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % (d['ms_per_step']*1e3))")
    echo "round $r  $v  $us us"
    if [ -z "${best[$v]}" ] || (( $(echo "$us < ${best[$v]}" | bc -l) )); then best[$v]=$us; fi
  done
done
for v in "${variants[@]}"; do echo "BEST  $v  ${best[$v]} us"; done
