#!/bin/bash
O=gpurun_out/r3_run4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q -s -k "linear_layers or whole_step or harness" > $O/pytest_r3.log 2>&1; echo "rc=$?" >> $O/pytest_r3.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py -x -q -k "linear or Linear or dx_column or c3 or c4" > $O/pytest_lin.log 2>&1; echo "rc=$?" >> $O/pytest_lin.log
{
echo "== new kernels (default)"; FFH_GEMM_CFG=-1 timeout 600 python3 tools/gemm_big.py child 32768x3456x1024 32768x1024x1024 32768x1024x512 32768x512x256 32768x256x128 4096x1024x1024 4096x3456x1024 2>&1 | grep -v "DLRM\|amdgpu.ids"
} > $O/gemm_big.txt 2>&1
timeout 400 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench.json
bash tools/r3_trace.sh > $O/trace.log 2>&1
tail -3 $O/pytest_r3.log; tail -3 $O/pytest_lin.log; cat $O/gemm_big.txt
python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['kernels']['linear_largest_layer']['fwd'], d['kernels']['linear_largest_layer']['bwd']['us'])"
cat gpurun_out/r3_trace/step_timeline.txt
