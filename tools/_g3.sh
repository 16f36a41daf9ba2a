cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_round2.py tests/test_bf16_mode.py tests/test_launchers.py -m gpu -q -x > $O/pytest_r2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2.log
grep -v "^\[DLRM\]" $O/pytest_r2.log | tail -30
