#!/bin/bash
# round 4, visit E: merged next-pass histogram in the sort (bit-exactness + microbenchmark), capture-exchange crash backtrace, whole GPU suite
R=$(pwd); O=$R/gpurun_out/r4_e; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3b.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -k "emb or sort or fuzz or fused" > $O/pytest_emb.log 2>&1; echo "pytest emb rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_emb.log | tail -2 | tee -a $O/summary.txt
python3 tools/microbench.py emb terabyte-4tables terabyte-26 terabyte-small-tables kaggle-rank-of-8 giant-colshard 2>&1 | grep -v amdgpu.ids | tee $O/emb.txt
python3 bench.py --workload kaggle --steps 20 --warmup 5 --force-exchange --force-graph "--shim-flags=--capture-exchange --backtrace-on-crash" --no-cpu-baseline --no-secondary > $O/capture.out 2> $O/capture.err; echo "capture rc=$?" | tee -a $O/summary.txt; tail -40 $O/capture.err | tee -a $O/summary.txt
timeout 3000 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round4.py::test_exchange_step_captured_as_a_graph_equals_eager_bit_for_bit > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest_all.log | tail -2 | tee -a $O/summary.txt
