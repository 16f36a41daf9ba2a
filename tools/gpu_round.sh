#!/bin/bash
# one GPU-box visit: the whole -m gpu suite, then the bench lines of the variants added last
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[DLRM\]" | tail -15 > gpurun_out/pytest_gpu.txt
tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --steps 300 --warmup 30 > gpurun_out/bench_kaggle.json 2> gpurun_out/bench_kaggle.err; tail -c 1500 gpurun_out/bench_kaggle.json
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --shim-flags "--zipf-alpha 1.05" > gpurun_out/bench_kaggle_zipf.json 2> gpurun_out/bench_kaggle_zipf.err; head -c 300 gpurun_out/bench_kaggle_zipf.json; echo
timeout 600 python bench.py --workload giant-row --force-exchange --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_giant_row.json 2> gpurun_out/bench_giant_row.err; head -c 300 gpurun_out/bench_giant_row.json; echo; tail -3 gpurun_out/bench_giant_row.err
timeout 600 python bench.py --workload giant --force-exchange --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_giant_col.json 2> gpurun_out/bench_giant_col.err; head -c 300 gpurun_out/bench_giant_col.json; echo; tail -3 gpurun_out/bench_giant_col.err
