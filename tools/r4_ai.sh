#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4_ai; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
export FFH_TOOLS_LIB=$R/tools/lab/libffhip_lab.so
for e in "X=1" "FFH_GLDS_DW_MAX=4e9"; do echo "== $e" | tee -a $O/out.txt; env $e python3 tools/dw_db_probe.py 8192x512x256 16384x512x256 8192x256x128 2>&1 | grep -E "^[0-9]|dW act relu premasked db|dW act relu db|dW act none db" | tee -a $O/out.txt; done
L="--backend tools/lab/libffhip_lab.so"
b() { env "$1" python3 bench.py --no-cpu-baseline --no-secondary "${@:2}" "--shim-flags=$L" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do for w in "--workload mlperf --steps 50 --warmup 5" "--per-gpu-batch 8192 --steps 60 --warmup 10"; do
echo "glds dw max 1e9 | $w | $(b X=1 $w)" | tee -a $O/out.txt
echo "glds dw max 4e9 | $w | $(b FFH_GLDS_DW_MAX=4e9 $w)" | tee -a $O/out.txt
done; done
