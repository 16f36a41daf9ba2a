#!/bin/bash
O=gpurun_out/r3_run7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_bf16_mode.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
tail -15 $O/pytest_bf16.log | cut -c1-300
for T in 0 1; do
  FFM_NO_BF16_TWINS=$([ $T = 0 ] && echo 1 || echo "") ; export FFM_NO_BF16_TWINS; [ $T = 1 ] && unset FFM_NO_BF16_TWINS
  timeout 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --shim-flags=--allow-tensor-op-math-conversion 2>/dev/null | grep '^{' | tail -1 > $O/bench_bf16_twins$T.json
  python3 -c "import json; d=json.load(open('$O/bench_bf16_twins$T.json')); print('twins $T', d['value'], d['ms_per_step'], d['kernels']['linear_largest_layer']['fwd'], d['kernels']['linear_largest_layer']['bwd']['us'])"
done
