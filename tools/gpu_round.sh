#!/bin/bash
# one GPU-box visit: the whole -m gpu suite, then bench lines of the variants added last
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[DLRM\]" | tail -15 > gpurun_out/pytest_gpu.txt
tail -3 gpurun_out/pytest_gpu.txt
for wl in mlperf mlperf-allpairs; do
  timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; head -c 400 gpurun_out/bench_$wl.json; echo; tail -2 gpurun_out/bench_$wl.err
done
