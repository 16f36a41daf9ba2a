#!/bin/bash
# A longer random sweep than the test-suite's (spare GPU minutes): embedding kernels bit for bit, Linear in all math modes (3 = the split mode on every shape, half of the cases with three-plane images).
cd "$(dirname "$0")/.."
N=${1:-150}
for seed in 101 102 103; do timeout 1500 python3 tools/fuzz_embedding.py $N $seed 2>&1 | tail -2; done
[ -n "$EMB_ONLY" ] || for mode in 0 1 2 3; do for seed in 201 202; do timeout 1500 python3 tools/fuzz_linear.py $N $seed $mode 2>&1 | tail -2; done; done
