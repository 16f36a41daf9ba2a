cd $GRAFT_REPO_ROOT
for B in 32768 4096; do
for v in 0 1 0 1; do
FFH_SPLITK_BALANCE=$v python3 bench.py --per-gpu-batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('B=$B balance=$v', d['value'], d['ms_per_step'], 'bwd', k['linear_largest_layer']['bwd']['us'])"
done; done
FFH_SPLITK_BALANCE=0 python3 tools/gemm_big.py -1 32768x3456x1024 4096x3456x1024 4096x1024x1024 8192x1024x1024 2>&1 | grep -v "DLRM\|amdgpu.ids\|CFG" | cut -c1-110
FFH_SPLITK_BALANCE=1 python3 tools/gemm_big.py -1 32768x3456x1024 4096x3456x1024 4096x1024x1024 8192x1024x1024 2>&1 | grep -v "DLRM\|amdgpu.ids\|CFG" | cut -c1-110
