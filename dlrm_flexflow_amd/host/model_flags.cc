// model_flags.cc -- FFConfig: defaults and the command line
// (one of the translation units of the host shim: model_internal.h lists them)
#include "model_internal.h"

// =============================================================================================
// FFConfig  [ref: src/runtime/model.cc:2211-2403]
// =============================================================================================
FFConfig::FFConfig() {
  epochs = 1;
  batchSize = 64;
  printFreq = 10;
  numNodes = 1;
  cpusPerNode = 0;
  workersPerNode = 0;
  learningRate = 0.01f;
  weightDecay = 0.0001f;
  workSpaceSize = (size_t)1 << 30;
  syntheticInput = false;
  profiling = false;
  perform_fusion = false;
  computationMode = COMP_MODE_TRAINING;
  device = 0;
  seed = 0;
  enable_graph = true;
  overlap_embedding = true;
  dense_embedding_update = false;
  force_exchange = false;
  parallel_dw = true;
  async_launch = false;   // measured on MI355X / ROCm 7.2: no gain over one issuing thread (280 vs 272 us per Kaggle step)
  column_shard_rows = 0;
  row_shard_rows = 0;
  replicate_embedding_rows = 0;
  fuse_loss = true;
  timing_events = false;
  attach_events = true;
  fuse_pair = true;
  mlp_chain = true;
  trace_mode = -1;
  bucket_allreduce = -1;
  allreduce_bucket_floats = 1 << 20;
  direct_allreduce = false;
  big_dw_chunks = 0;
  big_dw_min_weights = 2 << 20;
  mlp_chain_max_batch = 8192;
  mlp_chain_fwd_max_batch = 4096;
  // measured on whole steps (profiles/r05_ab_chain.txt) and alone (tools/chain_bench.py, profiles/r05_microbench_mlp_chain.txt): the
  // backward chain of the bottom MLPs beats the per-layer calls up to 4096 samples per GPU (alone 46 vs 86 us at 4096, 37 vs 74 at the
  // Kaggle shape; steps 1.163 vs 1.188 ms and 0.170 vs 0.187); at 8192 it is faster alone (72 vs 116) and -- since the table update takes
  // three launches at that size (the bucket form: the update no longer sits beside the whole bottom backward) -- on the step as well:
  // MLPerf shape 1.162-1.173 vs 1.177-1.179 ms, Terabyte shape at 8192 samples 2.077 vs 2.099 (before: 1.225 vs 1.200).  The forward chain
  // is level with the per-layer kernels at 4096 samples (28 vs 31 us alone, the step unchanged), slower below (28 vs 24 at
  // 2048) and above (8192: 2.085 vs 2.077 ms with it); a chain beyond ~200 K weights (the Kaggle top MLP: 352 K) loses below 4096 samples -- every CU streams every weight from
  // L2 for its 16 rows, which bounds these kernels
  mlp_chain_fwd_min_batch = 4096;
  mlp_chain_max_weights = 200000;
  dx_scatter = true;
  dx_colsum = true;
  early_sort = -1;
  pad_linear_k = true;
  capture_exchange = false;
  bf16_twins = true;
  bf16_convert_twins = true;
  bf16_exact_small_backward = true;
  force_async_launch = false;
  sparse_embedding_optimizer = false;
  allow_tensor_op_math_conversion = false;
  fp32_split_bf16x3 = false;
  deterministic = false;
  memset(&comm, 0, sizeof comm);
  comm.rank = 0;
  comm.world_size = 1;
}

void FFConfig::parse_args(char** argv, int argc) {
  for (int i = 1; i < argc; i++) {
    // "--flag value" as the reference's parser takes it, and "--flag=value" (one token: survives a launcher's word splitting)
    const char* eq = strncmp(argv[i], "--", 2) == 0 ? strchr(argv[i], '=') : nullptr;
    auto is = [&](const char* a) { return eq ? (strlen(a) == (size_t)(eq - argv[i]) && !strncmp(argv[i], a, (size_t)(eq - argv[i]))) : !strcmp(argv[i], a); };
    auto next = [&]() -> const char* {
      if (eq) return eq + 1;
      if (i + 1 >= argc) die("flag %s needs a value", argv[i]);
      return argv[++i];
    };
    if (is("-e") || is("--epochs")) { epochs = atoi(next()); continue; }
    if (is("-b") || is("--batch-size")) { batchSize = atoi(next()); continue; }
    if (is("--lr") || is("--learning-rate")) { learningRate = (float)atof(next()); continue; }
    if (is("--wd") || is("--weight-decay")) { weightDecay = (float)atof(next()); continue; }
    if (is("-p") || is("--print-freq")) { printFreq = atoi(next()); continue; }
    if (is("-d") || is("--dataset")) { dataset_path = next(); continue; }
    if (is("--import") || is("--import-strategy")) { import_strategy_file = next(); continue; }
    if (is("--export") || is("--export-strategy")) { export_strategy_file = next(); continue; }
    if (is("-ll:gpu")) { workersPerNode = atoi(next()); continue; }
    if (is("--nodes")) { numNodes = atoi(next()); continue; }
    if (is("-ll:cpu")) { cpusPerNode = atoi(next()); continue; }
    if (is("--profiling")) { profiling = true; continue; }
    if (is("--fusion")) { perform_fusion = true; continue; }
    // Legion/Realm pass-through flags of the reference scripts: accepted, meaningless here
    if (is("-ll:fsize") || is("-ll:zsize") || is("-ll:util") || is("-ll:csize") || is("--budget") || is("--search-budget") ||
        is("--alpha") || is("--search-alpha") || is("--simulator-workspace-size") || is("--strategy") ||
        is("--machine-model-version") || is("--machine-model-file") || is("--simulator-segment-size") ||
        is("--simulator-max-num-segments") || is("--taskgraph")) { next(); continue; }
    // cublasSetMathMode(CUBLAS_TENSOR_OP_MATH) on every handle [ref: src/runtime/model.cc:2282-2403, src/runtime/model.cu:81-83]
    if (is("--allow-tensor-op-math-conversion")) { allow_tensor_op_math_conversion = true; continue; }
    if (is("--fp32-split-bf16x3")) { fp32_split_bf16x3 = true; continue; }      // this build: fp32-accurate GEMMs on the bf16 pipe (ff_hip.h)
    if (is("-dm:memoize") || is("-dm:memorize") || is("--overlap") || is("--enable-parameter-parallel") ||
        is("--enable-attribute-parallel") || is("--enable-propagation")) continue;
    // this build
    if (is("--seed")) { seed = strtoull(next(), nullptr, 10); continue; }
    if (is("--deterministic")) { deterministic = true; continue; }
    if (is("--backend")) { backend_lib = next(); continue; }
    if (is("--device")) { device = atoi(next()); continue; }
    if (is("--no-trace")) { enable_graph = false; continue; }
    if (is("--no-overlap")) { overlap_embedding = false; continue; }
    if (is("--dense-embedding-update")) { dense_embedding_update = true; continue; }
    if (is("--force-exchange")) { force_exchange = true; continue; }
    if (is("--serial-dw")) { parallel_dw = false; continue; }
    if (is("--inline-launch")) { async_launch = false; continue; }
    if (is("--async-launch")) { async_launch = true; continue; }
    if (is("--column-shard-rows")) { column_shard_rows = atoll(next()); continue; }
    if (is("--row-shard-rows")) { row_shard_rows = atoll(next()); continue; }
    if (is("--replicate-embedding-rows")) { replicate_embedding_rows = atoll(next()); continue; }
    if (is("--no-fused-loss")) { fuse_loss = false; continue; }
    if (is("--timing-events")) { timing_events = true; continue; }
    if (is("--no-attach-event")) { attach_events = false; continue; }
    if (is("--no-fused-pair")) { fuse_pair = false; continue; }
    if (is("--no-mlp-chain")) { mlp_chain = false; continue; }
    if (is("--always-replay")) { trace_mode = 1; continue; }
    if (is("--adaptive-replay")) { trace_mode = 0; continue; }
    if (is("--bucket-allreduce")) { bucket_allreduce = 1; continue; }
    if (is("--no-bucket-allreduce")) { bucket_allreduce = 0; continue; }
    if (is("--allreduce-bucket-floats")) { allreduce_bucket_floats = atoll(next()); continue; }
    if (is("--direct-allreduce")) { direct_allreduce = true; continue; }
    if (is("--big-dw-chunks")) { big_dw_chunks = atoi(next()); continue; }
    if (is("--big-dw-min-weights")) { big_dw_min_weights = atoll(next()); continue; }
    if (is("--mlp-chain-max-batch")) { mlp_chain_max_batch = atoll(next()); continue; }
    if (is("--mlp-chain-fwd-min-batch")) { mlp_chain_fwd_min_batch = atoll(next()); continue; }
    if (is("--mlp-chain-fwd-max-batch")) { mlp_chain_fwd_max_batch = atoll(next()); continue; }
    if (is("--mlp-chain-max-weights")) { mlp_chain_max_weights = atoll(next()); continue; }
    if (is("--no-dx-scatter")) { dx_scatter = false; continue; }
    if (is("--no-dx-colsum")) { dx_colsum = false; continue; }
    if (is("--no-early-sort")) { early_sort = 0; continue; }
    if (is("--early-sort")) { early_sort = 1; continue; }
    if (is("--no-pad-linear-k")) { pad_linear_k = false; continue; }
    if (is("--capture-exchange")) { capture_exchange = true; continue; }
    if (is("--no-bf16-exact-small-backward")) { bf16_exact_small_backward = false; continue; }      // A/B: every wide layer's backward on the bf16 pipe
    if (is("--no-bf16-convert-twins")) { bf16_convert_twins = false; continue; }    // A/B: no twin by conversion behind an fp32-kernel layer
    if (is("--no-bf16-twins")) { bf16_twins = false; continue; }               // A/B and tests: tensor-op mode rounding its operands inside the kernels
    if (is("--force-async-launch")) { force_async_launch = true; continue; }   // tests: the launch-worker threads on a synchronous backend
    if (is("--sparse-embedding-optimizer")) { sparse_embedding_optimizer = true; continue; }
  }
}
