cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
FFH_GEMM_CFG=-1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/gemm_big.py child 32768x1024x1024 > $O/log.txt 2>&1
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $S | cut -c1-200
