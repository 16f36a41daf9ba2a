cd $GRAFT_REPO_ROOT
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('fp32', d['value'], d['ms_per_step']); 
for m in ('tensor_op_bf16_mode','fp32_split_bf16x3_mode'): print(m, json.dumps(k[m])[:1200])"
