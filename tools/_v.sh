cd $GRAFT_REPO_ROOT
python3 tools/longrun_check.py kaggle 2>&1 | grep -v "DLRM\|amdgpu" | tail -8
python3 tools/longrun_check.py mlperf 2>&1 | grep -v "DLRM\|amdgpu" | tail -8
for seed in 301 302; do timeout 900 python3 tools/fuzz_embedding.py 150 $seed 2>&1 | tail -1; done
for mode in 0 1 2; do timeout 900 python3 tools/fuzz_linear.py 100 4$mode $mode 2>&1 | tail -1; done
