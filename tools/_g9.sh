cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
FFH_GEMM_NO_PLR=$v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('NO_PLR=$v', d['value'], d['ms_per_step'], 'lin fwd', k['linear_largest_layer']['fwd'], 'bwd', k['linear_largest_layer']['bwd']['us'], k['linear_largest_layer']['bwd']['achieved'], 'step TF', k['whole_step_device']['mlp_tflops_over_whole_step'])"
done
