cd $GRAFT_REPO_ROOT
python3 tools/gemm_big.py -1,1,2 32768x512x256 32768x256x128 4096x512x256 4096x256x128 8192x512x256 2>&1 | grep -v "DLRM\|amdgpu.ids"
