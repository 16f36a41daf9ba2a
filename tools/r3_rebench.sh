#!/bin/bash
# the bench lines that quote profiles/r03_pmc_traffic.json, re-taken after that file was refreshed
O=gpurun_out/refresh; mkdir -p $O; P=r03
line() { grep '^{' | tail -1; }
python3 bench.py 2> $O/bench_default.err | line > $O/${P}_bench_terabyte.json
python3 bench.py --no-cpu-baseline --no-secondary --shim-flags=--allow-tensor-op-math-conversion 2>/dev/null | line > $O/${P}_bench_terabyte_bf16_mode.json
python3 bench.py --no-cpu-baseline --no-secondary --shim-flags=--fp32-split-bf16x3 2>/dev/null | line > $O/${P}_bench_terabyte_split_bf16x3_mode.json
python3 bench.py --force-exchange --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_terabyte_exchange_1rank.json
for f in $O/${P}_bench_terabyte*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic_source_current'))" 2>&1 | tail -1)"; done
cat $O/bench_default.err | tail -5
