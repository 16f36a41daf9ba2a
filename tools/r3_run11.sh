#!/bin/bash
O=gpurun_out/r3_run11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 400 python3 bench.py --workload kaggle --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/k_$name.json; python3 -c "import json; d=json.load(open('$O/k_$name.json')); print('$name', d['value'], d['ms_per_step'], d['config']['step_us_graph_vs_eager'])"; }
run default A=1
run thinrows_always FFH_THIN_ROWS_MIN_BATCH=1024
run default2 A=1
run no_sk FFH_GEMM_NO_SK=1
run cap4096 FFH_EMB_FWD_CAP=4096
