cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_launchers.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 900 python3 bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json; echo "bench rc=$?"
head -c 3000 $O/bench_default.json; echo
tail -5 $O/bench_default.err
bash tools/pmc_traffic.sh > $O/pmc.log 2>&1; tail -40 $O/pmc.log
