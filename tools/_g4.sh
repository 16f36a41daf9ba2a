cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
for mode in graph eager; do
  F=""; [ $mode = graph ] && F="--force-graph"; [ $mode = eager ] && F="--no-trace"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$mode -- python3 bench.py --workload kaggle --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $F 2> $O/$mode.err | grep '^{' > $O/kaggle_$mode.json
  T=$(find $O/prof_$mode -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/kaggle_${mode}_timeline.txt
  find $O/prof_$mode -name "*.csv" -size +10M -delete
done
python3 -c "
import json
for m in ('graph','eager'):
    d=json.load(open('$O/kaggle_%s.json'%m)); print(m, d['ms_per_step'], d['config']['step_us_graph_vs_eager'], d['config']['step_graph'])
"
cat $O/kaggle_graph_timeline.txt
cat $O/kaggle_eager_timeline.txt
