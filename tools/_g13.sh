cd $GRAFT_REPO_ROOT
for B in 32768 4096 2048; do
for v in 0 16384 0 16384; do
W=""; [ $B = 2048 ] && W="--workload kaggle"
FFH_GLDS_DW_KMAX=$v python3 bench.py $W --per-gpu-batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B KMAX=$v', d['value'], d['ms_per_step'])"
done; done
