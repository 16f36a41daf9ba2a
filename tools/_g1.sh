cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
python3 bench.py --workload terabyte --per-gpu-batch 32768 --steps 20 --warmup 3 --no-cpu-baseline 2> $O/tb32k.err | grep '^{' > $O/tb32k.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tb -- python3 bench.py --workload terabyte --per-gpu-batch 32768 --steps 10 --warmup 2 --no-cpu-baseline --no-trace 2> $O/prof_tb.err | grep '^{' > $O/tb32k_prof.json
S=$(find $O/prof_tb -name "*kernel_stats.csv" | head -1); cp $S $O/tb32k_kernel_stats.csv
T=$(find $O/prof_tb -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/tb32k_timeline.txt
find $O/prof_tb -name "*.csv" -size +10M -delete
python3 bench.py --workload mlperf --per-gpu-batch 8192 --steps 20 --warmup 3 --no-cpu-baseline 2> $O/ml.err | grep '^{' > $O/ml8k.json
head -c 1500 $O/tb32k.json; echo; head -30 $O/tb32k_kernel_stats.csv
