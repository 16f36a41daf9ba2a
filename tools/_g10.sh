cd $GRAFT_REPO_ROOT
echo "default"; python3 tools/gemm_big.py -1 32768x1024x1024 32768x1024x512 32768x512x256 4096x1024x1024 4096x1024x512 2>&1 | grep -v "DLRM\|amdgpu.ids\|CFG"
echo "FFH_GEMM_NO_GLDS=1"; FFH_GEMM_NO_GLDS=1 python3 tools/gemm_big.py -1 32768x1024x1024 32768x1024x512 32768x512x256 4096x1024x1024 4096x1024x512 2>&1 | grep -v "DLRM\|amdgpu.ids\|CFG"
