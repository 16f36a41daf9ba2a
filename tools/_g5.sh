cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tbgraph -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --force-graph 2> $O/tbgraph.err | grep '^{' > $O/tb_graph.json
T=$(find $O/prof_tbgraph -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T -3 > $O/tb_graph_timeline.txt
find $O/prof_tbgraph -name "*.csv" -size +10M -delete
cat $O/tb_graph_timeline.txt
