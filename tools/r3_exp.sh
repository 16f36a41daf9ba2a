timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "emb or sgd or fused" 2>&1 | tail -2
timeout 300 python tools/microbench.py emb 2>&1 | grep -E "terabyte"
timeout 300 python tools/microbench.py emb 2>&1 | grep -E "terabyte"
