cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "emb or fused or sgd or full_size or zipf or kaggle" 2>&1 | grep -v "^\[DLRM\]" | tail -4
python3 tools/microbench.py emb 2>&1 | grep -v amdgpu.ids
