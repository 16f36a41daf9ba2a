#!/bin/bash
# A/B of the lab library's environment switches on ONE box, interleaved rounds:
#   tools/ab.sh ROUNDS "<bench.py args>" "VAR1=1" "VAR2=1 VAR3=0" ...
# (tools/build_variant.sh tools/lab/libffhip_lab.so -DFFH_LAB first: the release library reads no environment variable.)  The first
# variant is always the lab library with every switch at its default.
rounds=$1; shift
args=$1; shift
declare -A best
variants=("FFH_NONE=0" "$@")
for r in $(seq $rounds); do
  for v in "${variants[@]}"; do
    us=$(env $v timeout 300 python3 bench.py --no-cpu-baseline --no-secondary $args --shim-flags="--backend tools/lab/libffhip_lab.so" 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % (d['ms_per_step']*1e3))")
    echo "round $r  $v  $us us"
    if [ -z "${best[$v]}" ] || python3 -c "import sys; sys.exit(0 if float('$us') < float('${best[$v]}') else 1)"; then best[$v]=$us; fi
  done
done
for v in "${variants[@]}"; do echo "BEST  $v  ${best[$v]} us"; done
