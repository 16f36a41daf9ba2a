cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --per-gpu-batch 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-trace 2>/dev/null | grep '^{' > $O/b4096.json
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/b4096_timeline.txt
find $O/prof -name "*.csv" -size +10M -delete
cat $O/b4096_timeline.txt
