cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; rm -rf $O; mkdir -p $O
for c in terabyte-4tables terabyte-26; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -- python3 tools/microbench.py emb $c > $O/$c.log 2>&1
S=$(find $O/$c -name "*kernel_stats.csv" | head -1); echo "== $c"; grep -E "emb_|radix" $S | awk -F'","' '{printf "%-70s calls %s avg %.1f us\n", substr($1,2,70), $2, $4/1000}'
done
