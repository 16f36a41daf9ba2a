#!/bin/bash
# one box visit: the whole GPU suite, the embedding microbenchmark, the default bench line
out=gpurun_out/full; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; grep -E "passed|failed|rror" $out/tests.txt | tail -3
timeout 600 python tools/microbench.py emb 2>&1 | grep -v amdgpu.ids | tee $out/microbench_emb.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
