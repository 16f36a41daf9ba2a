#!/bin/bash
O=gpurun_out/r3_run6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -c "
import torch
print('prio range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > $O/prio.txt 2>&1
run() { name=$1; shift; env "$@" timeout 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench_$name.json; python3 -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run base A=1
run side_low FFH_STREAM_PRIOS=0,1,0
run side_low_main_high FFH_STREAM_PRIOS=-1,1,-1
run side_high FFH_STREAM_PRIOS=0,-1,0
run fwdcap512 FFH_EMB_FWD_CAP=512
run fwdcap1024 FFH_EMB_FWD_CAP=1024
run side_low_cap512 FFH_STREAM_PRIOS=0,1,0 FFH_EMB_FWD_CAP=512
cat $O/prio.txt
