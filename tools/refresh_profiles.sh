#!/bin/bash
# Regenerates the judged artefacts under profiles/ on one GPU box (outputs land in gpurun_out/refresh, copied by hand).
R=$(pwd); O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 bench.py --probe 2> $O/bench_probe.err | grep '^{' > $O/r01_bench_kaggle.json
python3 bench.py 2> $O/bench_default.err | grep '^{' > $O/r01_bench_kaggle_default.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline 2> $O/prof.err | grep '^{' > $O/r01_bench_kaggle_under_rocprof.json
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/r01_bench_kaggle_kernel_stats.csv
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $T > $O/r01_bench_kaggle_step_timeline.txt
for k in emb_fwd_kernel emb_sgd_small_kernel "gemm_glds_kernel<false, false, 64" gemm_glds_bwd_kernel; do python3 tools/kernel_avg.py $T "$k"; done > $O/r01_bench_kaggle_probe_averages.txt
find $O/prof -name "*.csv" -size +10M -delete
python3 bench.py --no-cpu-baseline --shim-flags "--zipf-alpha 1.05" 2>/dev/null | grep '^{' > $O/r01_bench_kaggle_zipf.json
python3 bench.py --force-exchange --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r01_bench_kaggle_exchange_1rank.json
for wl in terabyte mlperf giant; do python3 bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r01_bench_$wl.json; done
python3 bench.py --workload giant-row --force-exchange --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r01_bench_giant_row.json
python3 bench.py --workload giant --force-exchange --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r01_bench_giant_col.json
for f in $O/*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])")"; done
cat $O/r01_bench_kaggle_step_timeline.txt $O/r01_bench_kaggle_probe_averages.txt
