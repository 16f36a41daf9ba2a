#!/bin/bash
# round 4, first box visit: where the per-rank step (4096 samples) goes -- kernel traces of the plain and the exchange-forced step,
# the B = 4096 / MLPerf GEMM shapes next to hipBLASLt, the fused update at the per-rank and the 26-table shape
R=$(pwd); O=$R/gpurun_out/r4_first; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
for v in plain exch; do
  F=""; [ $v = exch ] && F="--force-exchange"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary $F > $O/bench_$v.log 2>&1
  grep '^{' $O/bench_$v.log | tail -1 > $O/bench_b4096_$v.json
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_summary.py $T > $O/timeline_b4096_$v.txt 2>&1
  S=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats_b4096_$v.csv
  find $O/prof_$v -name "*.csv" -size +10M -delete
done
python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/bench_b4096_unprofiled.json
python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --force-exchange 2>/dev/null | grep '^{' | tail -1 > $O/bench_b4096_exch_unprofiled.json
python3 tools/gemm_big.py -1 4096x3456x1024 4096x1024x1024 4096x1024x512 4096x512x256 4096x256x128 8192x479x1024 8192x1024x1024 8192x1024x512 8192x512x256 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/gemm_b4096.txt
python3 tools/microbench.py emb terabyte-4tables terabyte-26 2>&1 | grep -v amdgpu.ids > $O/emb.txt
cat $O/timeline_b4096_plain.txt $O/gemm_b4096.txt $O/emb.txt
for f in $O/bench_b4096_*.json; do echo "$(basename $f): $(python3 -c "import json; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; done
