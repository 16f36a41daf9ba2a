#!/bin/bash
# ad-hoc: tests + bench of the tensor-op mode with the LDS-DMA kernel
mkdir -p gpurun_out/bf16a
timeout 1500 python -m pytest tests/test_bf16_mode.py -x -q -m gpu > gpurun_out/bf16a/tests.txt 2>&1; tail -5 gpurun_out/bf16a/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --shim-flags=--allow-tensor-op-math-conversion > gpurun_out/bf16a/bench_dma.json 2> gpurun_out/bf16a/bench_dma.err; tail -c 600 gpurun_out/bf16a/bench_dma.json
FFH_BF16_NO_DMA=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --shim-flags=--allow-tensor-op-math-conversion > gpurun_out/bf16a/bench_old.json 2> gpurun_out/bf16a/bench_old.err; tail -c 300 gpurun_out/bf16a/bench_old.json
timeout 300 python tools/bf16_twin_probe.py > gpurun_out/bf16a/twin_probe.txt 2>&1; tail -12 gpurun_out/bf16a/twin_probe.txt
