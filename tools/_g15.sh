cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-trace"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- $B > $O/f.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/h -- $B > $O/h.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/t -- $B > $O/t.log 2>&1
python3 tools/pmc_summary.py --all $(find $O -name "*counter_collection.csv") > $O/summary.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r2g/summary.json'))
for k,v in d.items():
    if 'gemm_f32_kernel<128' in k or 'emb_fwd' in k:
        print(k, {c:int(x['mean']) for c,x in v.items()})
PY
tail -3 $O/t.log
