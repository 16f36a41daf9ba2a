#!/bin/bash
O=gpurun_out/r3_run10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for C in 4096 1024 768 512 256; do
  FFH_EMB_FWD_CAP=$C timeout 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench_cap$C.json
  python3 -c "import json; d=json.load(open('$O/bench_cap$C.json')); r=d['roofline']; print('cap $C', d['value'], d['ms_per_step'], 'gather alone', r['us_per_launch'], 'in step', r.get('us_per_launch_in_step'), 'update', d['kernels']['embedding_bwd_sgd_fused']['us_per_launch'], d['kernels']['embedding_bwd_sgd_fused'].get('us_per_launch_in_step'))"
done
