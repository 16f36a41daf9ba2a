#!/bin/bash
O=gpurun_out/r3_run5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_model.py -x -q -k "linear or Linear or whole_step or golden or kaggle_shape_hip or config1" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for N in 64 256 512; do
  FFH_SKINNY_NBLK=$N timeout 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench_nblk$N.json
done
bash tools/r3_trace.sh > $O/trace.log 2>&1
tail -3 $O/pytest.log
for N in 64 256 512; do python3 -c "import json; d=json.load(open('$O/bench_nblk$N.json')); print('nblk $N', d['value'], d['ms_per_step'])"; done
cat gpurun_out/r3_trace/step_timeline.txt
