cd $GRAFT_REPO_ROOT
for v in 1e30 3e9 1e30 3e9; do
FFH_GLDS_DW_MAX=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('DW_MAX=$v', d['value'], d['ms_per_step'], 'step TF', k['whole_step_device']['mlp_tflops_over_whole_step'])"
done
for v in 1e30 3e9 1e30 3e9; do
FFH_GLDS_DW_MAX=$v python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('B4096 DW_MAX=$v', d['value'], d['ms_per_step'])"
done
