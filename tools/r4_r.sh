#!/bin/bash
# round 4 visit R: the Kaggle-shape step as a hipGraph of ONE stream (no fork / join edges) vs eager on three streams
R=$(pwd); O=$R/gpurun_out/r4_r; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
b() { python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'))"; }
for i in 1 2; do
for w in "--workload kaggle --steps 300 --warmup 30" "--workload tiny --steps 300 --warmup 30" "--workload giant --steps 200 --warmup 20"; do
echo "eager 3 streams      | $w | $(b $w --no-trace)" | tee -a $O/summary.txt
echo "graph 3 streams      | $w | $(b $w --force-graph)" | tee -a $O/summary.txt
echo "graph 1 stream       | $w | $(b $w --force-graph '--shim-flags=--no-overlap --serial-dw')" | tee -a $O/summary.txt
echo "graph no-overlap     | $w | $(b $w --force-graph '--shim-flags=--no-overlap')" | tee -a $O/summary.txt
echo "graph serial-dw      | $w | $(b $w --force-graph '--shim-flags=--serial-dw')" | tee -a $O/summary.txt
echo "eager 1 stream       | $w | $(b $w --no-trace '--shim-flags=--no-overlap --serial-dw')" | tee -a $O/summary.txt
done
done
