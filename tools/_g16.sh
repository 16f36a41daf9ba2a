cd $GRAFT_REPO_ROOT
for i in 1 2; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('occ4', d['value'], d['ms_per_step'], 'lin fwd', k['linear_largest_layer']['fwd']['us'], 'bwd', k['linear_largest_layer']['bwd']['us'], 'step TF', k['whole_step_device']['mlp_tflops_over_whole_step'])"
done
python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B4096', d['value'], d['ms_per_step'])"
