cd $GRAFT_REPO_ROOT
FFH_GEMM_NO_PLR=1 python3 tools/gemm_big.py -1 32768x3456x1024 32768x1024x1024 4096x3456x1024 4096x1024x1024 2>&1 | grep -v "DLRM\|amdgpu.ids"
python3 tools/gemm_big.py -1 32768x3456x1024 32768x1024x1024 4096x3456x1024 4096x1024x1024 2>&1 | grep -v "DLRM\|amdgpu.ids"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -k "linear or c3 or c4 or bmm" 2>&1 | grep -v "^\[DLRM\]" | tail -4
