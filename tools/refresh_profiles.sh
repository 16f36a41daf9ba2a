#!/bin/bash
# Regenerates the judged artefacts under profiles/ on one GPU box.  Outputs land in gpurun_out/refresh/ with their final
# names (r06_*); copy them into profiles/ afterwards.   gpurun --timeout 3300 -- 'bash tools/refresh_profiles.sh'
R=$(pwd); O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
P=${P:-r06}
line() { grep '^{' | tail -1; }

# 0. HBM traffic of the embedding kernels first (two counter-only passes): the bench lines below quote it, and warn when the file on
#    disk names kernels the library no longer has
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; cp gpurun_out/pmc_traffic/pmc_traffic.json $O/${P}_pmc_traffic.json; cp $O/${P}_pmc_traffic.json profiles/${P}_pmc_traffic.json

# 1. the default command, unprofiled: headline + roofline + kernels (bf16 mode, Kaggle secondary) + cpu_baseline
python3 bench.py 2> $O/bench_default.err | line > $O/${P}_bench_terabyte.json

# 2. the same command under the kernel trace (program directly after --; single-shape: no secondary blocks)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-secondary 2> $O/prof.err | line > $O/${P}_bench_terabyte_under_rocprof.json
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/${P}_bench_terabyte_kernel_stats.csv
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $T > $O/${P}_bench_terabyte_step_timeline.txt
for k in emb_fwd_kernel emb_sgd_reduce_kernel radix_scatter_kernel "gemm_sk_kernel<false, false, 0" "gemm_sk_kernel<false, true, 1" "gemm_sk_kernel<true, true, 3, true" "gemm_sk_kernel<true, true, 3, false"; do
  python3 tools/kernel_avg.py $T "$k"
done > $O/${P}_bench_terabyte_probe_averages.txt
find $O/prof -name "*.csv" -size +10M -delete

# 3. SQ counters of the step's kernels
bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq/summary.json $O/${P}_pmc_sq_counters_step.json
bash tools/pmc_gemm.sh > $O/pmc_gemm.log 2>&1; cp gpurun_out/pmc_gemm/summary.json $O/${P}_pmc_sq_counters.json

# 4. the other workloads (one GPU)
python3 bench.py --no-cpu-baseline --no-secondary --shim-flags=--allow-tensor-op-math-conversion 2>/dev/null | line > $O/${P}_bench_terabyte_bf16_mode.json
python3 bench.py --no-cpu-baseline --no-secondary --shim-flags=--fp32-split-bf16x3 2>/dev/null | line > $O/${P}_bench_terabyte_split_bf16x3_mode.json
python3 bench.py --workload kaggle --steps 300 --warmup 30 2>/dev/null | line > $O/${P}_bench_kaggle.json
python3 bench.py --workload tiny --steps 500 --warmup 50 2>/dev/null | line > $O/${P}_bench_tiny.json
python3 bench.py --workload mlperf --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_mlperf.json
python3 bench.py --workload giant --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_giant.json
python3 bench.py --force-exchange --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_terabyte_exchange_1rank.json
python3 bench.py --workload kaggle --force-exchange --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_kaggle_exchange_1rank.json
python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line > $O/${P}_bench_terabyte_b4096.json
python3 bench.py --per-gpu-batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --force-exchange 2>/dev/null | line > $O/${P}_bench_terabyte_b4096_exchange_1rank.json
# ... and its kernel timelines (the per-rank step of the 8-GPU job), plain and exchange-forced
for v in plain exchange; do
  F=""; [ $v = exchange ] && F="--force-exchange"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b4096_$v -- python3 bench.py --per-gpu-batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary $F > /dev/null 2>&1
  T=$(find $O/prof_b4096_$v -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/${P}_terabyte_b4096_${v}_step_timeline.txt
  find $O/prof_b4096_$v -name "*.csv" -size +10M -delete
done

timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_mlperf -- python3 bench.py --workload mlperf --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
T=$(find $O/prof_mlperf -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/${P}_mlperf_step_timeline.txt; find $O/prof_mlperf -name "*.csv" -size +10M -delete

# 5. hipGraph replay against eager launches, one step each
for mode in graph eager; do
  F="--force-graph"; [ $mode = eager ] && F="--no-trace"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$mode -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary $F 2>/dev/null | line > $O/tb_$mode.json
  T=$(find $O/prof_$mode -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T -3 > $O/${P}_terabyte_${mode}_step_timeline.txt
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/profk_$mode -- python3 bench.py --workload kaggle --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $F 2>/dev/null | line > $O/kg_$mode.json
  T=$(find $O/profk_$mode -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $T > $O/${P}_kaggle_${mode}_step_timeline.txt
  find $O/prof_$mode $O/profk_$mode -name "*.csv" -size +10M -delete
done

# 6. GEMM microbenchmarks: fp32 kernels next to hipBLASLt (torch.mm) in one process; fp32 vs tensor-op (bf16) mode; the lab
python3 tools/gemm_big.py -1 32768x3456x1024 32768x1024x1024 32768x1024x512 4096x3456x1024 4096x1024x1024 4096x1024x512 4096x512x256 8192x512x1024 8192x479x1024 8192x1024x1024 8192x1024x512 8192x1024x256 8192x512x256 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/${P}_microbench_gemm_vs_hipblaslt.txt
python3 tools/gemm_bf16_bench.py 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/${P}_microbench_gemm_bf16_mode.txt
# the lab binary is built here from its source (never committed)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_big_lab.hip -lrocblas -o tools/lab/gemm_big_lab 2> $O/lab_build.err && timeout 300 tools/lab/gemm_big_lab 32768 1024 3456 > $O/${P}_lab_gemm_big.txt 2>&1
python3 tools/microbench.py emb > $O/${P}_microbench_embedding.txt 2>&1
# the chain launches of narrow Linear layers next to the per-layer calls (round 5), alone
{ python3 tools/chain_bench.py 2048 4096 8192; python3 tools/chain_bench.py 4096 --single; } 2>&1 | grep "^B=" > $O/${P}_microbench_mlp_chain.txt
# the stand-alone lab of the persistent fp32 GEMM (built here from its source), with its ablation modes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_sk_lab.hip -o tools/lab/gemm_sk_lab 2>> $O/lab_build.err && { timeout 120 tools/lab/gemm_sk_lab 32768 1024 1024 1 256 1; timeout 120 tools/lab/gemm_sk_lab 32768 3456 1024 1 256 0; } > $O/${P}_lab_gemm_sk.txt 2>&1
# tensor-op mode: one big layer with and without bf16 twins
{ python3 tools/bf16_twin_probe.py 32768x3456x1024; python3 tools/bf16_twin_probe.py 32768x1024x1024; } 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/${P}_microbench_bf16_twins.txt
# SQ counters of the split-bf16x3 forward GEMM alone (MFMA busy cycles against GRBM_GUI_ACTIVE: what bounds that kernel)
bash tools/pmc_x3.sh 2 > $O/pmc_x3.log 2>&1; cp gpurun_out/pmc_x3/summary.json $O/${P}_pmc_split_bf16x3_gemm.json
# round 6: the split mode from three-plane images -- one layer through the C-ABI with and without images, the stand-alone lab, step timelines
python3 tools/x3_image_probe.py 32768x3456x1024 32768x1024x1024 32768x1024x512 8192x3456x1024 4096x3456x1024 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/${P}_microbench_x3_images.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_x3_lab.hip -o tools/lab/gemm_x3_lab 2>> $O/lab_build.err && { timeout 300 tools/lab/gemm_x3_lab 32768 3456 1024 1; timeout 300 tools/lab/gemm_x3_lab 32768 1024 1024; } 2>&1 | grep -v "split3" > $O/${P}_lab_gemm_x3.txt
tools/step_timeline.sh ${P}_terabyte_split --fp32-split-bf16x3 > /dev/null 2>&1; cp gpurun_out/${P}_terabyte_split_step_timeline.txt $O/
tools/step_timeline.sh ${P}_terabyte_bf16 --allow-tensor-op-math-conversion > /dev/null 2>&1; cp gpurun_out/${P}_terabyte_bf16_step_timeline.txt $O/
tools/step_timeline.sh ${P}_terabyte_b4096_split --fp32-split-bf16x3 --per-gpu-batch 4096 > /dev/null 2>&1; cp gpurun_out/${P}_terabyte_b4096_split_step_timeline.txt $O/

for f in $O/${P}_bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))" 2>&1 | tail -1)"; done
cat $O/${P}_bench_terabyte_probe_averages.txt
ls -la $O | head -60
