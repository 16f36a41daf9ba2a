#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the embedding kernels at the default bench shape: two counter-only passes (no trace domains).
R=$(pwd); O=$R/gpurun_out/pmc_traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-trace"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- $B > $O/f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- $B > $O/w.log 2>&1
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py terabyte $F $W 879230976 1315438592 > $O/pmc_traffic.json
find $O -name "*.csv" -size +20M -delete
cat $O/pmc_traffic.json | head -60
