#!/bin/bash
# One parametrised GPU-box visit:   gpurun --timeout T -- 'bash tools/visit.sh <name> <step> [<step> ...]'
# Every step writes under gpurun_out/<name>/ and appends one line to gpurun_out/<name>/summary.txt.  Steps:
#   smoke            __graft_entry__.smoke()
#   suite            the whole GPU suite (pytest -m gpu)
#   t:<expr>         pytest -m gpu -k '<expr>'            (e.g. t:mlp_chain)
#   f:<file>         pytest -m gpu tests/<file>
#   bench:<tag>:<args>        one bench.py line (no cpu baseline, no secondary), e.g. bench:b4096:--per-gpu-batch_4096_--steps_100_--warmup_10
#                             (underscores stand for spaces; a literal underscore is written as %)
#   ab:<tag>:<argsA>:<argsB>  the two bench lines interleaved three times (boxes and minutes differ by a few percent)
#   tl:<tag>:<args>  rocprofv3 kernel trace of a short bench run + the one-step timeline (tools/trace_summary.py)
#   py:<tag>:<script>_<args>  python3 tools/<script> <args>
#   refresh          tools/refresh_profiles.sh (every judged artefact under profiles/)
R=$(pwd); N=$1; shift; O=$R/gpurun_out/$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
un() { echo "$1" | sed 's/_/ /g; s/%/_/g'; }
line() { grep '^{' | tail -1; }
short() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'samples/s', d['ms_per_step'], 'ms/step')" 2>/dev/null; }
for step in "$@"; do
  case "$step" in
    smoke) python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt ;;
    suite) timeout 3300 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "suite rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)" | tee -a $O/summary.txt ;;
    t:*) K="${step#t:}"; timeout 1800 python3 -m pytest tests -m gpu -x -q -k "$K" > "$O/pytest_$K.log" 2>&1; echo "pytest -k $K rc=$? $(grep -E 'passed|failed' "$O/pytest_$K.log" | tail -1)" | tee -a $O/summary.txt; grep -E "^E |Error" "$O/pytest_$K.log" | head -20 | tee -a $O/summary.txt ;;
    f:*) F="${step#f:}"; timeout 1800 python3 -m pytest tests/$F -m gpu -x -q > "$O/pytest_$F.log" 2>&1; echo "pytest $F rc=$? $(grep -E 'passed|failed' "$O/pytest_$F.log" | tail -1)" | tee -a $O/summary.txt; grep -E "^E |Error" "$O/pytest_$F.log" | head -20 | tee -a $O/summary.txt ;;
    bench:*) IFS=: read -r _ T A <<< "$step"; python3 bench.py --no-cpu-baseline --no-secondary $(un "$A") 2> $O/bench_$T.err | line > $O/bench_$T.json; echo "bench $T: $(short < $O/bench_$T.json)" | tee -a $O/summary.txt ;;
    ab:*) IFS=: read -r _ T A B <<< "$step"
      for i in 1 2 3; do
        a=$(python3 bench.py --no-cpu-baseline --no-secondary $(un "$A") 2>/dev/null | line | short)
        b=$(python3 bench.py --no-cpu-baseline --no-secondary $(un "$B") 2>/dev/null | line | short)
        echo "ab $T #$i: A[$(un "$A")] $a | B[$(un "$B")] $b" | tee -a $O/summary.txt
      done ;;
    tl:*) IFS=: read -r _ T A <<< "$step"
      timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$T -- python3 bench.py --no-cpu-baseline --no-secondary $(un "$A") > /dev/null 2>&1
      C=$(find $O/prof_$T -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $C > $O/timeline_$T.txt 2>&1
      find $O/prof_$T -name "*.csv" -size +10M -delete; echo "timeline $T: $(head -1 $O/timeline_$T.txt)" | tee -a $O/summary.txt ;;
    py:*) IFS=: read -r _ T A <<< "$step"; timeout 1500 python3 tools/$(un "$A") > $O/$T.txt 2>&1; echo "py $T rc=$?" | tee -a $O/summary.txt; grep -v "amdgpu.ids\|^\[DLRM\]" $O/$T.txt | tail -40 ;;
    refresh) bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?" | tee -a $O/summary.txt; tail -25 $O/refresh.log ;;
    *) echo "unknown step $step" | tee -a $O/summary.txt ;;
  esac
done
echo "---- summary"; cat $O/summary.txt
