#!/bin/bash
# round 4 visit Q: the default bench line again (in-step probes now also when the timed steps are graph replays)
R=$(pwd); O=$R/gpurun_out/r4_q; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
python3 bench.py 2> $O/bench.err | grep '^{' | tail -1 > $O/r04_bench_terabyte.json; head -c 400 $O/r04_bench_terabyte.json
