#!/bin/bash
# one box visit for the fused-update work: bit-exactness (parity tests that touch the embedding kernels, a fuzz seed), then the microbenchmark
out=gpurun_out/emb; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3b.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q -k "emb or fused or sort or sgd or giant or table" > $out/tests.txt 2>&1
grep -E "passed|failed|rror" $out/tests.txt | tail -5
timeout 600 python tools/fuzz_embedding.py 60 777 2>&1 | tail -1
timeout 600 python tools/microbench.py emb 2>&1 | grep -v amdgpu.ids | tee $out/microbench.txt
