#!/bin/bash
# round-3 GPU visit 1: new parity tests + persistent-grid A/B of the fp32 GEMMs
O=gpurun_out/r3_run1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_integration_stubs.py "tests/test_gpu_model.py::test_replicated_tables_steady_state_is_ordered_behind_the_slab_update" -x -q -s > $O/pytest_new.log 2>&1
echo "pytest new rc=$?" >> $O/pytest_new.log
for P in 0 2 3 4; do
  echo "== FFH_GEMM_PERSIST=$P" >> $O/gemm_persist.txt
  FFH_GEMM_PERSIST=$P FFH_GEMM_CFG=$([ $P = 0 ] && echo -1 || echo -2) timeout 300 python3 tools/gemm_big.py child 32768x3456x1024 32768x1024x1024 32768x1024x512 4096x1024x1024 2>&1 | grep -v "DLRM\|amdgpu.ids" >> $O/gemm_persist.txt
done
for P in 0 3; do
  echo "== dW through the register-staged kernel (FFH_GLDS_DW_KMAX=1000000), FFH_GEMM_PERSIST=$P" >> $O/gemm_persist.txt
  FFH_GLDS_DW_KMAX=1000000 FFH_GEMM_PERSIST=$P FFH_GEMM_CFG=-2 timeout 300 python3 tools/gemm_big.py child 32768x3456x1024 32768x1024x1024 32768x1024x512 2>&1 | grep -v "DLRM\|amdgpu.ids" >> $O/gemm_persist.txt
done
for P in 0 3; do
  FFH_GEMM_PERSIST=$P timeout 400 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench_persist$P.json
done
tail -5 $O/pytest_new.log; cat $O/gemm_persist.txt
for P in 0 3; do python3 -c "import json; d=json.load(open('$O/bench_persist$P.json')); print('persist $P', d['value'], d['ms_per_step'], d['kernels']['linear_largest_layer'])"; done
