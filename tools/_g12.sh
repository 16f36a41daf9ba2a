cd $GRAFT_REPO_ROOT
for B in 4096 8192 16384; do
for v in 1e30 3e9 1e9 1e30 3e9 1e9; do
FFH_GLDS_DW_MAX=$v python3 bench.py --per-gpu-batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B DW_MAX=$v', d['value'], d['ms_per_step'])"
done; done
