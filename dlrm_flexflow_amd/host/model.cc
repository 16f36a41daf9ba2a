// model.cc -- FFModel / Op / Optimizer / Initializer of the reference API over the kernel C-ABI.
// See ffmodel.h for the reference declarations each class mirrors.
#include "ffmodel.h"

#include <algorithm>
#include <set>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ffh_rng.h"
#include "backend.h"

namespace {

[[noreturn]] void die(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "FATAL: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  abort();   // the reference asserts/exits on every such condition [ref: include/cuda_helper.h:6-47]
}

double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

size_t dtype_size(DataType t) {
  switch (t) {
    case DT_FLOAT: return 4;
    case DT_DOUBLE: return 8;
    case DT_INT32: return 4;
    case DT_INT64: return 8;
    case DT_BOOLEAN: return 1;
    default: return 0;
  }
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace

// =============================================================================================
// LaunchWorker
// =============================================================================================
LaunchWorker::LaunchWorker(const KernelApi* _api, int _device) : api(_api), device(_device), wctx(nullptr), stop(false), busy(false) {
  th = std::thread([this] { run(); });
  drain();   // the ctx exists once the first (empty) round trip is over
}
LaunchWorker::~LaunchWorker() {
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  if (th.joinable()) th.join();
  if (wctx) api->ffh_ctx_destroy(wctx);
}
void LaunchWorker::run() {
  {
    ffh_ctx* c = nullptr;
    if (api->ffh_ctx_create(&c, device) != FFH_OK) die("launch worker: ffh_ctx_create(device %d) failed", device);   // also binds the thread to the device
    std::lock_guard<std::mutex> lk(mu);
    wctx = c;
  }
  cv_idle.notify_all();
  for (;;) {
    std::function<void(ffh_ctx*)> fn;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return stop || !q.empty(); });
      if (q.empty()) { if (stop) return; continue; }
      fn = std::move(q.front());
      q.pop_front();
      busy = true;
    }
    fn(wctx);
    {
      std::lock_guard<std::mutex> lk(mu);
      busy = false;
    }
    cv_idle.notify_all();
  }
}
void LaunchWorker::post(std::function<void(ffh_ctx*)> fn) {
  {
    std::lock_guard<std::mutex> lk(mu);
    q.push_back(std::move(fn));
  }
  cv.notify_one();
}
void LaunchWorker::drain() {
  std::unique_lock<std::mutex> lk(mu);
  cv_idle.wait(lk, [this] { return q.empty() && !busy && wctx != nullptr; });
}

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
    if (is("--no-bf16-convert-twins")) { bf16_convert_twins = false; continue; }    // A/B: no twin by conversion behind an fp32-kernel layer
    if (is("--no-bf16-twins")) { bf16_twins = false; continue; }               // A/B and tests: tensor-op mode rounding its operands inside the kernels
    if (is("--force-async-launch")) { force_async_launch = true; continue; }   // tests: the launch-worker threads on a synchronous backend
    if (is("--sparse-embedding-optimizer")) { sparse_embedding_optimizer = true; continue; }
  }
}

// =============================================================================================
// Tensor / Parameter
// =============================================================================================
Tensor::Tensor() : numDim(0), data_type(DT_FLOAT), sync_type(NONE), owner_op(nullptr), owner_idx(0), impl(nullptr) {
  for (int i = 0; i < MAX_TENSOR_DIM; i++) adim[i] = 0;
}

size_t Tensor::get_volume() const {
  size_t v = 1;
  for (int i = 0; i < numDim; i++) v *= (size_t)adim[i];
  return v;
}

int64_t Tensor::rows() const {
  int64_t r = 1;
  for (int i = 1; i < numDim; i++) r *= adim[i];
  return r;
}

namespace {
int64_t local_rows(const Tensor& t, const FFModel*) { return t.impl ? t.impl->rows_local : 0; }
}  // namespace

template <typename T>
bool Tensor::set_tensor(const FFModel* model, const std::vector<int>& dims, const T* data) {
  if (!impl || !impl->ptr) die("set_tensor before compile()");
  if (sizeof(T) != dtype_size(data_type)) die("set_tensor: element type does not match the tensor's data type");
  size_t vol = 1;
  for (int d : dims) vol *= (size_t)d;
  const int64_t cols_ = adim[0];
  const int64_t nrows = (int64_t)(vol / (size_t)cols_);
  if (vol % (size_t)cols_ != 0 || nrows != impl->rows_local)
    die("set_tensor: %zu elements given, this rank holds %lld x %lld", vol, (long long)impl->rows_local, (long long)cols_);
  // the fused table update (side stream) still reads the sparse ids and writes the tables, a forked weight-gradient GEMM
  // (dw stream) still reads activations: a host write must land behind both, as every host read does (copy_out)
  if (model->dw_worker) model->dw_worker->drain();
  if (model->side_worker) model->side_worker->drain();
  model->check(model->api->ffh_stream_sync(model->ctx, model->side_stream), "set_tensor sync");
  model->check(model->api->ffh_stream_sync(model->ctx, model->dw_stream), "set_tensor sync");
  model->check(model->api->ffh_stream_sync(model->ctx, model->ar_stream), "set_tensor sync");
  if (impl->ld == cols_) {
    model->check(model->api->ffh_memcpy_h2d(model->ctx, impl->ptr, data, vol * sizeof(T), model->stream), "set_tensor");
  } else {
    for (int64_t r = 0; r < nrows; r++)
      model->check(model->api->ffh_memcpy_h2d(model->ctx, (char*)impl->ptr + r * impl->ld * sizeof(T), data + r * cols_,
                                               cols_ * sizeof(T), model->stream), "set_tensor");
  }
  model->check(model->api->ffh_stream_sync(model->ctx, model->stream), "set_tensor sync");
  model->note_weight_write(impl->ptr);       // tensor-op mode: the weights' bf16 twin is reconverted before the next step
  return true;
}

namespace {
template <typename T>
bool copy_out(const FFModel* model, const Tensor& t, const void* base, int64_t ld, T* data) {
  if (!base) die("get_tensor before compile() (or tensor has no gradient)");
  const int64_t cols_ = t.adim[0];
  const int64_t nrows = local_rows(t, model);
  if (model->dw_worker) model->dw_worker->drain();
  if (model->side_worker) model->side_worker->drain();
  model->check(model->api->ffh_stream_sync(model->ctx, model->stream), "get_tensor sync");
  model->check(model->api->ffh_stream_sync(model->ctx, model->side_stream), "get_tensor sync");
  model->check(model->api->ffh_stream_sync(model->ctx, model->dw_stream), "get_tensor sync");
  model->check(model->api->ffh_stream_sync(model->ctx, model->ar_stream), "get_tensor sync");
  if (ld == cols_) {
    model->check(model->api->ffh_memcpy_d2h(model->ctx, data, base, (size_t)nrows * cols_ * sizeof(T), model->stream), "get_tensor");
  } else {
    for (int64_t r = 0; r < nrows; r++)
      model->check(model->api->ffh_memcpy_d2h(model->ctx, data + r * cols_, (const char*)base + r * ld * sizeof(T),
                                               cols_ * sizeof(T), model->stream), "get_tensor");
  }
  model->check(model->api->ffh_stream_sync(model->ctx, model->stream), "get_tensor sync");
  return true;
}
}  // namespace

template <typename T>
bool Tensor::get_tensor(const FFModel* model, T* data) const {
  if (sizeof(T) != dtype_size(data_type)) die("get_tensor: element type does not match the tensor's data type");
  return copy_out<T>(model, *this, impl ? impl->ptr : nullptr, impl ? impl->ld : 0, data);
}
template <typename T>
bool Tensor::get_grad(const FFModel* model, T* data) const {
  return copy_out<T>(model, *this, impl ? impl->grad : nullptr, impl ? impl->grad_ld : 0, data);
}
template <typename T>
bool Parameter::set_weights(const FFModel* model, const std::vector<int>& dims, const T* data) {
  return set_tensor<T>(model, dims, data);
}
template <typename T>
bool Parameter::get_weights(const FFModel* model, T* data) const {
  return get_tensor<T>(model, data);
}
template bool Tensor::set_tensor<float>(const FFModel*, const std::vector<int>&, const float*);
template bool Tensor::set_tensor<int64_t>(const FFModel*, const std::vector<int>&, const int64_t*);
template bool Tensor::get_tensor<float>(const FFModel*, float*) const;
template bool Tensor::get_tensor<int64_t>(const FFModel*, int64_t*) const;
template bool Tensor::get_grad<float>(const FFModel*, float*) const;
template bool Parameter::set_weights<float>(const FFModel*, const std::vector<int>&, const float*);
template bool Parameter::get_weights<float>(const FFModel*, float*) const;

// =============================================================================================
// Initializers [ref: src/runtime/initializer.cc, initializer_kernel.cu:24-276]
// The reference draws from cuRAND; streams here come from include/ffh_rng.h.
// =============================================================================================
void ZeroInitializer::init(const FFModel* ff, const Parameter* p) {
  ff->check(ff->api->ffh_zero(ff->ctx, p->impl->ptr, p->get_volume() * sizeof(float), ff->stream), "ZeroInitializer");
  ff->note_weight_write(p->impl->ptr);
}
void ConstantInitializer::init(const FFModel* ff, const Parameter* p) {
  ff->check(ff->api->ffh_fill_f32(ff->ctx, (float*)p->impl->ptr, (int64_t)p->get_volume(), value, ff->stream), "ConstantInitializer");
  ff->note_weight_write(p->impl->ptr);
}
void UniformInitializer::init(const FFModel* ff, const Parameter* p) {
  ff->check(ff->api->ffh_init_uniform(ff->ctx, (float*)p->impl->ptr, (int64_t)p->get_volume(),
                                      ff->config.seed * 0x9E3779B1ULL + (uint64_t)(uint32_t)seed, min_val, max_val, ff->stream),
            "UniformInitializer");
  ff->note_weight_write(p->impl->ptr);
}
void NormInitializer::init(const FFModel* ff, const Parameter* p) {
  // Box-Muller on the host from the counter-based stream, then one upload (MLP tensors are small)
  const size_t n = p->get_volume();
  std::vector<float> h(n);
  const uint64_t s = ff->config.seed * 0x9E3779B1ULL + (uint64_t)(uint32_t)seed;
  for (size_t i = 0; i < n; i++) {
    const uint64_t h1 = ffh_hash(s, 2 * i), h2 = ffh_hash(s, 2 * i + 1);
    const double u1 = ((double)(h1 >> 40) + 1.0) / 16777216.0;   // (0, 1]
    const double u2 = (double)(h2 >> 40) / 16777216.0;           // [0, 1)
    const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
    h[i] = (float)((double)mean + (double)stddev * z);
  }
  ff->check(ff->api->ffh_memcpy_h2d(ff->ctx, p->impl->ptr, h.data(), n * sizeof(float), ff->stream), "NormInitializer");
  ff->check(ff->api->ffh_stream_sync(ff->ctx, ff->stream), "NormInitializer sync");
  ff->note_weight_write(p->impl->ptr);
}
void GlorotUniform::init(const FFModel* ff, const Parameter* p) {
  // scale = sqrt(6 / (fan_in + fan_out)) [ref: src/runtime/initializer_kernel.cu:24-60]
  const float fan = (float)(p->adim[0] + (p->numDim > 1 ? p->adim[1] : 0));
  const float scale = std::sqrt(6.0f / fan);
  ff->check(ff->api->ffh_init_uniform(ff->ctx, (float*)p->impl->ptr, (int64_t)p->get_volume(),
                                      ff->config.seed * 0x9E3779B1ULL + (uint64_t)(uint32_t)seed, -scale, scale, ff->stream),
            "GlorotUniform");
  ff->note_weight_write(p->impl->ptr);
}

// =============================================================================================
// PerfMetrics [ref: src/metrics_functions/metrics_functions.cc:20-80]
// =============================================================================================
PerfMetrics::PerfMetrics()
    : train_all(0), train_correct(0), cce_loss(0), sparse_cce_loss(0), mse_loss(0), rmse_loss(0), mae_loss(0) {
  start_time = now_us();
}
void PerfMetrics::update(const PerfMetrics& o) {
  train_all += o.train_all; train_correct += o.train_correct; cce_loss += o.cce_loss;
  sparse_cce_loss += o.sparse_cce_loss; mse_loss += o.mse_loss; rmse_loss += o.rmse_loss; mae_loss += o.mae_loss;
}
void PerfMetrics::print(int flags) const {
  std::string out = "[Metrics]";
  if (flags & 1) {
    const float acc = train_all ? train_correct * 100.0f / train_all : 0.0f;
    out += " accuracy: " + std::to_string(acc) + "% (" + std::to_string(train_correct) + " / " + std::to_string(train_all) + ")";
  }
  if (flags & 2) out += " mean_squared_error: " + std::to_string(train_all ? mse_loss / train_all : 0.0f);
  if (flags & 4) out += " root_mean_squared_error: " + std::to_string(train_all ? rmse_loss / train_all : 0.0f);
  if (flags & 8) out += " mean_absolute_error: " + std::to_string(train_all ? mae_loss / train_all : 0.0f);
  fprintf(stderr, "%s\n", out.c_str());
}

// =============================================================================================
// Op base
// =============================================================================================
std::string FFModel::get_operator_type_name(OperatorType type) const {
  switch (type) {
    case OP_LINEAR: return "Dense";
    case OP_EMBEDDING: return "Embedding";
    case OP_CONCAT: return "Concat";
    case OP_BATCHMATMUL: return "BatchMatmul";
    case OP_TRANSPOSE: return "Transpose";
    case OP_RESHAPE: return "Reshape";
    case OP_FLAT: return "Flat";
    case OP_TRIL: return "Tril";
    case OP_DOT_INTERACTION: return "DotInteraction";
    default: return "Unknown";
  }
}

Op::Op(FFModel& model, OperatorType type, const char* _name, int num_inputs, const Tensor* _inputs)
    : op_type(type), numInputs(num_inputs), numWeights(0), numOutputs(1), layer_index(-1), profiling(model.config.profiling) {
  // "<Type>_<guid>" from guid 100 [ref: src/runtime/model.cc:253-268]
  std::string pc = (_name == nullptr ? model.get_operator_type_name(type) : std::string(_name)) + "_" +
                   std::to_string(model.op_global_guid++);
  if (pc.length() >= MAX_OPNAME) die("operator name too long: %s", pc.c_str());
  strcpy(name, pc.c_str());
  if (num_inputs > MAX_NUM_INPUTS) die("%s: more than %d inputs", name, MAX_NUM_INPUTS);
  for (int i = 0; i < num_inputs; i++) inputs[i] = _inputs[i];
  outputs[0].owner_op = this;
  outputs[0].owner_idx = 0;
  outputs[0].data_type = DT_FLOAT;
  outputs[0].impl = new TensorImpl();
  model.tensor_impls.push_back(outputs[0].impl);
  outputs[0].impl->guid = (int)model.tensor_impls.size() - 1;
}

void Op::print_layer(const FFModel&) const {
  printf("%s: inputs %d ->", name, numInputs);
  for (int d = outputs[0].numDim - 1; d >= 0; d--) printf(" %d", outputs[0].adim[d]);
  printf("\n");
}

// =============================================================================================
// FFModel: construction, graph building
// =============================================================================================
// --profiling [ref: src/runtime/model.cc:2358-2362]: every op is bracketed by two events and waited for, as the reference's
// tasks do (src/ops/linear.cu:525-546) -- so nothing may overlap or fuse across ops while it is on
static FFConfig& profiling_schedule(FFConfig& c) {
  if (c.profiling) {
    c.overlap_embedding = false; c.enable_graph = false; c.parallel_dw = false; c.async_launch = false;
    c.fuse_pair = false; c.mlp_chain = false; c.attach_events = false; c.dx_scatter = false; c.timing_events = true;
  }
  return c;
}

FFModel::FFModel(FFConfig& _config)
    : op_global_guid(100), config(profiling_schedule(_config)), optimizer(nullptr), loss_type(LOSS_MEAN_SQUARED_ERROR_AVG_REDUCE),
      metrics_flags(0), seq_length(-1), api(nullptr), ctx(nullptr), stream(nullptr), side_stream(nullptr),
      ev_fork(nullptr), ev_join(nullptr), ev_grad_ready(nullptr), ev_update_done(nullptr), compiled(false),
      emb_forward_issued(false), emb_forward_joined(false), emb_update_pending(false), emb_sorted_early(false), mlp_weights(nullptr), mlp_grads(nullptr), mlp_count(0),
      act_slab(nullptr), act_grad_slab(nullptr), act_grad_bytes(0), workspace(nullptr), workspace_bytes(0), repl_workspace(nullptr), repl_workspace_bytes(0), d_perf(nullptr),
      xsend(nullptr), xrecv(nullptr), gsend(nullptr), grecv(nullptr), capturing_trace(-1), replaying_trace(-1), inputs_dirty(true), fork_recorded(false) {
  seed_counter = 0;
  dw_stream = nullptr; ev_dw_done = nullptr; big_dw_layer = -1; need_zero_act_grads = true; need_zero_gsend = true; dw_forked = false; mlp_grads_clean = false; dw_worker = side_worker = nullptr;
  rank = config.comm.world_size > 1 ? config.comm.rank : 0;
  world_size = config.comm.world_size > 1 ? config.comm.world_size : 1;
  if (world_size == 1 && config.workersPerNode > 1)
    die("-ll:gpu %d: one process per GPU and no communicator was supplied; use the `dlrm` binary (it starts its own %d ranks),\n"
        "       python dlrm_flexflow_amd/run_dlrm.py <flags>, or bench.py --gpus %d",
        config.workersPerNode, config.workersPerNode, config.workersPerNode);
  if (world_size > 1 && (!config.comm.alltoall_f32 || !config.comm.allreduce_sum_f32))
    die("world_size %d needs the ffcomm callbacks", world_size);
  if (config.batchSize % world_size != 0) die("batch size %d is not divisible by %d ranks", config.batchSize, world_size);
  local_batch = config.batchSize / world_size;
  exchange = world_size > 1 || (config.force_exchange && config.comm.alltoall_f32 && config.comm.allreduce_sum_f32);
  api = load_kernel_api(config.backend_lib);
  int rc = api->ffh_ctx_create(&ctx, config.device);
  if (rc != FFH_OK || !ctx) die("ffh_ctx_create(device %d) failed with %d on %s -- no usable GPU?", config.device, rc, api->path.c_str());
  if (config.allow_tensor_op_math_conversion) check(api->ffh_ctx_set_math_mode(ctx, FFH_MATH_TENSOR_OP_BF16), "tensor-op math mode");
  else if (config.fp32_split_bf16x3) check(api->ffh_ctx_set_math_mode(ctx, FFH_MATH_FP32_SPLIT_BF16X3), "split-bf16x3 math mode");
  if (config.deterministic) { check(api->ffh_ctx_set_deterministic(ctx, 1), "deterministic mode"); config.async_launch = false; }
  check(api->ffh_stream_create(ctx, &stream), "stream create");
  // the compute stream runs the Linear layers: the library's scratch for their stream-K / last-arriver forms is reserved here, once,
  // outside any capture (ffh_ctx_reserve_scratch, ABI 12: compute entry points never allocate)
  check(api->ffh_ctx_reserve_scratch(ctx, stream), "reserve scratch");
  check(api->ffh_stream_create(ctx, &side_stream), "stream create");
  check(api->ffh_stream_create(ctx, &dw_stream), "stream create");
  // (the row-block weight gradients of the biggest layer launch on dw_stream as their primary stream: its scratch too -- round-5 advisor)
  check(api->ffh_ctx_reserve_scratch(ctx, dw_stream), "reserve scratch");
  // the gradient buckets' stream (a fifth stream would share a hardware queue with one of the others: HIP maps streams onto four)
  check(api->ffh_stream_create(ctx, &ar_stream), "stream create");
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_dw_done), "event create");
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_z_free), "event create");
  z_reader_layer = -1; z_free_recorded = false;
  dw_worker = side_worker = nullptr;
  if (config.async_launch && (std::string(api->ffh_backend_name()).rfind("hip", 0) == 0 || config.force_async_launch)) {
    // asynchronous devices only: on the CPU oracle a "launch" is the computation itself
    dw_worker = new LaunchWorker(api, config.device);
    side_worker = new LaunchWorker(api, config.device);
    const int mm = config.allow_tensor_op_math_conversion ? FFH_MATH_TENSOR_OP_BF16 : (config.fp32_split_bf16x3 ? FFH_MATH_FP32_SPLIT_BF16X3 : FFH_MATH_DEFAULT);
    if (mm != FFH_MATH_DEFAULT) {
      check(api->ffh_ctx_set_math_mode(dw_worker->ctx(), mm), "math mode");
      check(api->ffh_ctx_set_math_mode(side_worker->ctx(), mm), "math mode");
    }
  }
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_fork), "event create");
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_join), "event create");
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_grad_ready), "event create");
  check((config.timing_events ? api->ffh_event_create : api->ffh_event_create_sync)(ctx, &ev_update_done), "event create");
}

FFModel::~FFModel() {
  if (!ctx) return;
  delete dw_worker; delete side_worker;
  api->ffh_device_sync(ctx);
  for (ffh_event e : layer_events) api->ffh_event_destroy(ctx, e);
  for (auto& kv : graphs) api->ffh_graph_destroy(ctx, kv.second);
  // device memory is released with the context's process; explicit frees keep long-lived hosts clean
  for (TensorImpl* t : tensor_impls) {
    if (t->ptr && !t->alias && t->bytes) api->ffh_free(ctx, t->ptr);
    delete t;
  }
  for (void* p : {w_twin, act_twin, grad_twin, (void*)ar_scratch}) if (p) api->ffh_free(ctx, p);
  for (void* p : {(void*)mlp_weights, (void*)mlp_grads, (void*)act_slab, (void*)act_grad_slab, workspace, repl_workspace, (void*)d_perf, (void*)xsend,
                  (void*)xrecv, (void*)gsend, (void*)grecv})
    if (p) api->ffh_free(ctx, p);
  for (Embedding* e : embeddings)
    for (void* p : {(void*)e->local_idx, (void*)e->partial, (void*)e->gfull, (void*)e->opt_state[0], (void*)e->opt_state[1]})
      if (p) api->ffh_free(ctx, p);
  for (Op* op : layers)
    if (op->op_type == OP_LINEAR && static_cast<Linear*>(op)->dx_map) api->ffh_free(ctx, static_cast<Linear*>(op)->dx_map);
  api->ffh_event_destroy(ctx, ev_fork); api->ffh_event_destroy(ctx, ev_join);
  api->ffh_event_destroy(ctx, ev_grad_ready); api->ffh_event_destroy(ctx, ev_update_done);
  api->ffh_event_destroy(ctx, ev_dw_done);
  api->ffh_event_destroy(ctx, ev_z_free);
  for (auto& kv : trace_tune) for (ffh_event& e : kv.second.ev) if (e) { api->ffh_event_destroy(ctx, e); e = nullptr; }
  for (ffh_event& e : probe_ev) if (e) { api->ffh_event_destroy(ctx, e); e = nullptr; }
  api->ffh_stream_destroy(ctx, stream); api->ffh_stream_destroy(ctx, side_stream); api->ffh_stream_destroy(ctx, dw_stream); api->ffh_stream_destroy(ctx, ar_stream);
  for (GradBucket& b : grad_buckets) { api->ffh_event_destroy(ctx, b.ready); api->ffh_event_destroy(ctx, b.ready_dw); api->ffh_event_destroy(ctx, b.done); }
  api->ffh_ctx_destroy(ctx);
  for (Op* op : layers) delete op;
  for (Initializer* i : owned_initializers) delete i;
  for (Tensor* t : input_tensors) delete t;
}

void FFModel::check(int rc, const char* what) const {
  if (rc != FFH_OK) die("%s failed (%d): %s", what, rc, api->ffh_last_error_string(ctx));
}

void* FFModel::dmalloc(size_t bytes) const {
  void* p = nullptr;
  int rc = api->ffh_malloc(ctx, &p, bytes);
  if (rc != FFH_OK || !p) die("device allocation of %zu bytes failed: %s", bytes, api->ffh_last_error_string(ctx));
  return p;
}

template <int NDIM>
Tensor FFModel::create_tensor(const int dims[], DataType data_type, const Op* owner_op, bool create_grad) {
  (void)create_grad;
  Tensor t;
  t.numDim = NDIM;
  t.data_type = data_type;
  t.owner_op = const_cast<Op*>(owner_op);
  for (int i = 0; i < NDIM; i++) t.adim[i] = dims[NDIM - 1 - i];   // Legion order [ref: src/runtime/model.cc:865-868]
  t.impl = new TensorImpl();
  tensor_impls.push_back(t.impl);
  t.impl->guid = (int)tensor_impls.size() - 1;
  t.impl->is_input = owner_op == nullptr;
  if (owner_op == nullptr) {
    Tensor* keep = new Tensor(t);
    input_tensors.push_back(keep);
  }
  return t;
}
template Tensor FFModel::create_tensor<1>(const int[], DataType, const Op*, bool);
template Tensor FFModel::create_tensor<2>(const int[], DataType, const Op*, bool);
template Tensor FFModel::create_tensor<3>(const int[], DataType, const Op*, bool);
template Tensor FFModel::create_tensor<4>(const int[], DataType, const Op*, bool);

template <int NDIM>
Parameter FFModel::create_weight(const int dims[], const Op* op, DataType data_type, Initializer* initializer, bool) {
  Parameter p;
  p.numDim = NDIM;
  p.data_type = data_type;
  p.owner_op = const_cast<Op*>(op);
  p.sync_type = world_size > 1 ? NCCL : PS;
  for (int i = 0; i < NDIM; i++) p.adim[i] = dims[NDIM - 1 - i];
  p.impl = new TensorImpl();
  tensor_impls.push_back(p.impl);
  p.impl->guid = (int)tensor_impls.size() - 1;
  (void)initializer;
  return p;
}
template Parameter FFModel::create_weight<1>(const int[], const Op*, DataType, Initializer*, bool);
template Parameter FFModel::create_weight<2>(const int[], const Op*, DataType, Initializer*, bool);

Tensor FFModel::dense(const Tensor& input, int outDim, ActiMode activation, bool use_bias, const Op* shared_op,
                      Initializer* kernel_initializer, Initializer* bias_initializer, const char* name) {
  // default initialisers [ref: src/ops/linear.cu:19-39]
  if (kernel_initializer == nullptr) kernel_initializer = new GlorotUniform(next_seed());
  if (bias_initializer == nullptr) bias_initializer = new ZeroInitializer();
  Linear* li = new Linear(*this, input, outDim, activation, use_bias, shared_op, kernel_initializer, bias_initializer, name);
  li->layer_index = (int)layers.size();
  layers.push_back(li);
  return li->outputs[0];
}

Tensor FFModel::embedding(const Tensor& input, int num_entries, int outDim, AggrMode aggr, const Op* shared_op,
                          Initializer* kernel_initializer, const char* name) {
  if (kernel_initializer == nullptr) kernel_initializer = new GlorotUniform(next_seed());   // [ref: src/ops/embedding.cu:19-34]
  Embedding* e = new Embedding(*this, input, num_entries, outDim, aggr, shared_op, kernel_initializer, name);
  e->layer_index = (int)layers.size();
  layers.push_back(e);
  return e->outputs[0];
}

Tensor FFModel::concat(int n, const Tensor* tensors, int axis, const char* name) {
  Concat* c = new Concat(*this, n, tensors, axis, name);
  c->layer_index = (int)layers.size();
  layers.push_back(c);
  return c->outputs[0];
}

Tensor FFModel::batch_matmul(const Tensor& A, const Tensor& B, int a_seq_length_dim, int b_seq_length_dim) {
  BatchMatmul* b = new BatchMatmul(*this, A, B, a_seq_length_dim, b_seq_length_dim);
  b->layer_index = (int)layers.size();
  layers.push_back(b);
  return b->outputs[0];
}

// =============================================================================================
// Linear [ref: src/ops/linear.cu]
// =============================================================================================
Linear::Linear(FFModel& model, const Tensor& input, int out_dim, ActiMode _activation, bool _use_bias, const Op* shared_op,
               Initializer* ki, Initializer* bi, const char* name)
    : Op(model, OP_LINEAR, name, 1, &input), in_channels(input.adim[0]), out_channels(out_dim), in_padded(input.adim[0]), activation(_activation),
      use_bias(_use_bias), discard_input_grad(input.owner_op == nullptr), dx_overwrite(false), dx_map(nullptr), dx_map_concat(nullptr), pair_upper(nullptr), fwd_done_by_pair(false), pair_lower(nullptr), dx_mask_by_x(false), dy_premasked(false), colsum_lower(nullptr), db_from_upper(false), fwd_done_by_chain(false),
      kernel_initializer(ki), bias_initializer(bi) {
  if (shared_op) die("%s: weight sharing is not supported on this path", this->name);
  if (input.data_type != DT_FLOAT) die("%s: input must be DT_FLOAT", this->name);
  if (activation != AC_MODE_NONE && activation != AC_MODE_RELU && activation != AC_MODE_SIGMOID && activation != AC_MODE_GELU)
    die("%s: activation %d not supported (NONE, RELU, SIGMOID, GELU)", this->name, (int)activation);
  outputs[0].numDim = input.numDim;
  for (int i = 1; i < input.numDim; i++) outputs[0].adim[i] = input.adim[i];
  outputs[0].adim[0] = out_dim;   // [ref: src/ops/linear.cu:41-71]
  numWeights = use_bias ? 2 : 1;
}
void Linear::create_output_and_partition(FFModel&) {}
void Linear::create_weights(FFModel& model) {
  const int kdims[2] = {out_channels, in_channels};
  weights[0] = model.create_weight<2>(kdims, this, DT_FLOAT, kernel_initializer);
  if (use_bias) {
    const int bdims[1] = {out_channels};
    weights[1] = model.create_weight<1>(bdims, this, DT_FLOAT, bias_initializer);
  }
}
void Linear::forward(const FFModel& ff) {
  const Tensor& x = inputs[0];
  const Tensor& y = outputs[0];
  const int64_t b = local_rows(y, &ff);
  if (fwd_done_by_pair) { fwd_done_by_pair = false; return; }      // the layer below computed this output in its launch
  if (fwd_done_by_chain) { fwd_done_by_chain = false; return; }    // ... or the lowest layer of its chain did
  if (!chain_fwd.empty() && ff.mlp_chain_usable(b, true)) {
    const int rc = ff.run_chain_fwd(this);
    if (rc == FFH_OK) { for (size_t i = 1; i < chain_fwd.size(); i++) chain_fwd[i]->fwd_done_by_chain = true; return; }
    if (rc != FFH_ERR_UNSUPPORTED) ff.check(rc, name);
    chain_fwd.clear();                                             // not a chain the library serves: the per-layer calls from now on
  }
  if (pair_upper) {
    Linear* up = pair_upper;
    const Tensor& yu = up->outputs[0];
    const int rc = ff.api->ffh_linear_pair_fwd(ff.ctx, (const float*)x.impl->ptr, x.impl->ld, (const float*)weights[0].impl->ptr,
                                               use_bias ? (const float*)weights[1].impl->ptr : nullptr, in_channels, (int)activation, (float*)y.impl->ptr,
                                               y.impl->ld, out_channels, (const float*)up->weights[0].impl->ptr,
                                               up->use_bias ? (const float*)up->weights[1].impl->ptr : nullptr, up->out_channels, (int)up->activation,
                                               (float*)yu.impl->ptr, yu.impl->ld, b, ff.stream);
    if (rc == FFH_OK) { up->fwd_done_by_pair = true; return; }
    if (rc != FFH_ERR_UNSUPPORTED) ff.check(rc, name);
    pair_upper = nullptr;                                          // not a shape the pair launch serves
  }
  // (in_padded: the layer as the kernel library sees it -- see allocate(), step 4a; equal to in_channels unless the input was padded)
  ff.check(ff.api->ffh_linear_fwd(ff.ctx, (const float*)x.impl->ptr, x.impl->ld, (float*)y.impl->ptr, y.impl->ld,
                                  (const float*)weights[0].impl->ptr, use_bias ? (const float*)weights[1].impl->ptr : nullptr,
                                  in_padded, out_channels, b, (int)activation, ff.stream), name);
  // tensor-op mode: this layer runs on the fp32 kernels (in_dim or out_dim below FFH_BF16_MIN_DIM) but feeds one on the bf16 pipe: its
  // output's twin by an explicit conversion (allocate(), step 7), so that the consumer reads both operands at two bytes per element
  if (out_twin && out_twin_x3) ff.check(ff.api->ffh_convert_f32_to_bf16x3(ff.ctx, (const float*)y.impl->ptr, b, out_channels, y.impl->ld, ff.stream), name);
  else if (out_twin) ff.check(ff.api->ffh_convert_f32_to_bf16(ff.ctx, out_twin, (const float*)y.impl->ptr, b * (int64_t)y.impl->ld, ff.stream), name);
}
int Linear::backward_pair(const FFModel& ff) {
  Linear* lo = pair_lower;
  const Tensor &xu = inputs[0], &yu = outputs[0], &xl = lo->inputs[0];
  const int64_t b = local_rows(yu, &ff);
  const int flags_u = dy_premasked ? FFH_LINEAR_DY_PREMASKED : 0;
  const int flags_l = (lo->dx_overwrite ? FFH_LINEAR_DX_OVERWRITE : 0) | (lo->dx_mask_by_x ? FFH_LINEAR_DX_MASK_BY_X : 0);
  const int rc = ff.api->ffh_linear_pair_bwd(
      ff.ctx, (const float*)xu.impl->ptr, xu.impl->ld, (const float*)yu.impl->ptr, yu.impl->ld, yu.impl->grad, yu.impl->grad_ld,
      (const float*)weights[0].impl->ptr, weights[0].impl->grad, (use_bias && !db_from_upper) ? weights[1].impl->grad : nullptr, in_channels, out_channels,
      (int)activation, flags_u, (const float*)xl.impl->ptr, xl.impl->ld, xl.impl->grad, xl.impl->grad_ld, xu.impl->grad, xu.impl->grad_ld,
      (const float*)lo->weights[0].impl->ptr, lo->in_channels, (int)lo->activation, flags_l, b, ff.stream);
  if (rc != FFH_OK) return rc;
  db_from_upper = false;        // (honoured above and consumed: the layer above may have produced this layer's bias gradient -- round-4 advisor)
  // what is left of the lower layer: dW / db over the whole batch, from the gradient the launch above wrote (premasked)
  ff.check(ff.api->ffh_linear_bwd_ex(ff.ctx, (const float*)xl.impl->ptr, xl.impl->ld, nullptr, xl.impl->grad_ld, (const float*)xu.impl->ptr, xu.impl->ld,
                                     xu.impl->grad, xu.impl->grad_ld, (const float*)lo->weights[0].impl->ptr, lo->weights[0].impl->grad,
                                     lo->use_bias ? lo->weights[1].impl->grad : nullptr, lo->in_channels, lo->out_channels, b, (int)lo->activation,
                                     FFH_LINEAR_ONLY_DW | FFH_LINEAR_DY_PREMASKED, ff.stream, nullptr), lo->name);
  return FFH_OK;
}
void Linear::backward(const FFModel& ff) { backward_part(ff, 0); }

// part 0: the whole backward; 1: the data gradient only (FFH_LINEAR_ONLY_DX on the compute stream) -- for FFModel::backward's row-block
// weight gradient of the biggest layer (backward_dw_rows), which follows it.
void Linear::backward_part(const FFModel& ff, int part) {
  // [ref: src/ops/linear.cu:632-635: "only support relu and sigmoid for now" -- an assert there, a named error here]
  if (activation == AC_MODE_GELU) die("%s: GELU has no backward (forward / inference only, as in the reference)", name);
  const Tensor& x = inputs[0];
  const Tensor& y = outputs[0];
  const int64_t b = local_rows(y, &ff);
  float* dx = discard_input_grad ? nullptr : x.impl->grad;
  // the weight-gradient GEMM gets its own stream only where it is long enough to pay for the fork and the join
  // (two event records + two waits, each a packet the command processor has to retire): measured on the Kaggle shape,
  // forking the 432x512 / 512x256 layers gains 22 us per step, forking the 256x64 / 64x16 / 13x512 ones loses 6
  const double macs = (double)in_channels * out_channels * (double)b;
  const bool fork = ff.config.parallel_dw && macs >= 1.0e8;
  const int flags = (dx_overwrite ? FFH_LINEAR_DX_OVERWRITE : 0) | (dx_mask_by_x ? FFH_LINEAR_DX_MASK_BY_X : 0) | (dy_premasked ? FFH_LINEAR_DY_PREMASKED : 0);
  const float *xp = (const float*)x.impl->ptr, *yp = (const float*)y.impl->ptr, *wp = (const float*)weights[0].impl->ptr;
  float *dyp = y.impl->grad, *dwp = weights[0].impl->grad, *dbp = (use_bias && !db_from_upper) ? weights[1].impl->grad : nullptr;
  const int64_t ldx = x.impl->ld, lddx = x.impl->grad_ld, ldy = y.impl->ld, lddy = y.impl->grad_ld;
  if (part != 1) db_from_upper = false;      // consumed by the call below that produces dW / db (part 1 is followed by a part 2 / 3 of this layer)
  // the lower layer's bias gradient rides on this call's data-gradient kernel where the library takes it
  struct ColsumScope {
    const FFModel& ff; Linear* lo;
    ColsumScope(const FFModel& f, Linear* l, bool produces_dx) : ff(f), lo(produces_dx ? l : nullptr) {
      if (lo) ff.check(ff.api->ffh_linear_bwd_set_dx_colsum(ff.ctx, lo->weights[1].impl->grad, lo->out_channels), "dx colsum");
    }
    ~ColsumScope() { if (lo) lo->db_from_upper = ff.api->ffh_linear_dx_colsum_used(ff.ctx) != 0; }
  } colsum_scope(ff, colsum_lower, dx != nullptr && (part == 0 || part == 1) && !ff.use_workers() && !ff.config.profiling);
  // split mode, this layer on the fp32 kernels under one that streams images: the image of the data gradient just stored (allocate(), step 7)
  auto image_dx = [&] { if (dx_image && dx) ff.check(ff.api->ffh_convert_f32_to_bf16x3(ff.ctx, dx, b, in_channels, lddx, ff.stream), name); };
  if (part == 1) {
    ff.check(ff.api->ffh_linear_bwd_ex(ff.ctx, xp, ldx, dx, lddx, yp, ldy, dyp, lddy, wp, dwp, dbp, in_padded, out_channels, b, (int)activation,
                                       flags | FFH_LINEAR_ONLY_DX, ff.stream, nullptr), name);
    image_dx();
    return;
  }
  if (fork && ff.use_workers()) {
    // two host threads: this one keeps walking the dX chain, the dW GEMM is issued by the dw worker on its stream
    const KernelApi* api = ff.api;
    ffh_stream dws = ff.dw_stream;
    ffh_event ev = ff.layer_events[layer_index];
    const int in = in_padded, out = out_channels, act = (int)activation;
    const char* nm = name;
    auto dw_call = [=](ffh_ctx* wc) {
      int rc = api->ffh_stream_wait_event(wc, dws, ev);
      if (rc == FFH_OK) rc = api->ffh_linear_bwd_ex(wc, xp, ldx, nullptr, lddx, yp, ldy, dyp, lddy, wp, dwp, dbp, in, out, b, act, flags | FFH_LINEAR_ONLY_DW, dws, nullptr);
      if (rc != FFH_OK) die("%s (weight gradient) failed (%d): %s", nm, rc, api->ffh_last_error_string(wc));
    };
    const bool sig = activation == AC_MODE_SIGMOID;   // its in-place activation pass (in the ONLY_DX call) must precede the dW GEMM
    if (!sig) { ff.check(api->ffh_event_record(ff.ctx, ev, ff.stream), "event"); ff.dw_worker->post(dw_call); }
    ff.check(api->ffh_linear_bwd_ex(ff.ctx, xp, ldx, dx, lddx, yp, ldy, dyp, lddy, wp, dwp, dbp, in, out, b, act, flags | FFH_LINEAR_ONLY_DX, ff.stream, nullptr), name);
    if (sig) { ff.check(api->ffh_event_record(ff.ctx, ev, ff.stream), "event"); ff.dw_worker->post(dw_call); }
    ff.dw_forked = true;
    return;
  }
  ffh_stream dws = ff.dw_stream;
  ff.check(ff.api->ffh_linear_bwd_ex(ff.ctx, xp, ldx, dx, lddx, yp, ldy, dyp, lddy, wp, dwp, dbp, in_padded, out_channels, b, (int)activation,
                                     flags, ff.stream, fork ? dws : nullptr), name);
  image_dx();
  if (fork) { ff.dw_forked = true; ff.dw1_used = true; }
}

// One row block of this layer's weight gradient: dW[row0 .. row0 + nrows)[:] (and db of the same rows) as an ONLY_DW call of its own on
// the weight-gradient stream, from the column slice of dy those outputs own.  For the bucketed all-reduce (FFModel::backward): the
// biggest layer's gradients are two thirds of the slab and complete last; cut into row blocks, the sum of block c over the ranks runs
// beside the GEMM of block c + 1.  The caller has issued the data gradient (part 1) and recorded layer_events[layer_index] behind it.
void Linear::backward_dw_rows(const FFModel& ff, int row0, int nrows) {
  const Tensor& x = inputs[0];
  const Tensor& y = outputs[0];
  const int64_t b = local_rows(y, &ff);
  const float *xp = (const float*)x.impl->ptr, *yp = (const float*)y.impl->ptr + row0;
  float *dyp = y.impl->grad + row0, *dwp = weights[0].impl->grad + (size_t)row0 * (size_t)weights[0].impl->ld;
  const float* wp = (const float*)weights[0].impl->ptr + (size_t)row0 * (size_t)weights[0].impl->ld;
  float* dbp = (use_bias && !db_from_upper) ? weights[1].impl->grad + row0 : nullptr;
  const int flags = FFH_LINEAR_ONLY_DW | FFH_LINEAR_DY_PREMASKED;       // (dy final: the caller checked)
  ff.check(ff.api->ffh_linear_bwd_ex(ff.ctx, xp, x.impl->ld, nullptr, x.impl->grad_ld, yp, y.impl->ld, dyp, y.impl->grad_ld, wp, dwp, dbp, in_padded, nrows, b,
                                     (int)activation, flags, ff.dw_stream, nullptr), name);
  ff.dw_forked = true;
  ff.dw1_used = true;
  ff.dw_stream_used_directly = true;
}

// ---- chains of narrow Linear layers (ffh_mlp_chain_fwd / _bwd, ABI 12; built in FFModel::allocate step 4e) -------------------
bool FFModel::mlp_chain_usable(int64_t rows, bool fwd) const {
  return config.mlp_chain && !config.profiling && !use_workers() && !config.allow_tensor_op_math_conversion &&
         !config.fp32_split_bf16x3 && rows <= config.mlp_chain_max_batch && (!fwd || (rows >= config.mlp_chain_fwd_min_batch && rows <= config.mlp_chain_fwd_max_batch));
}
static void fill_chain(const std::vector<Linear*>& ch, ffh_chain_layer* out) {
  for (size_t i = 0; i < ch.size(); i++) {
    const Linear* li = ch[i];
    ffh_chain_layer& d = out[i];
    d.w = (const float*)li->weights[0].impl->ptr;
    d.bias = li->use_bias ? (const float*)li->weights[1].impl->ptr : nullptr;
    d.y = (float*)li->outputs[0].impl->ptr; d.ldy = li->outputs[0].impl->ld;
    d.dy = li->outputs[0].impl->grad; d.lddy = li->outputs[0].impl->grad_ld;
    d.dw = li->weights[0].impl->grad;
    d.db = (li->use_bias && !li->db_from_upper) ? li->weights[1].impl->grad : nullptr;
    d.ldw = (int)li->weights[0].impl->ld; d.in_dim = li->in_padded; d.out_dim = li->out_channels; d.activation = (int)li->activation;
  }
}
int FFModel::run_chain_fwd(const Linear* lowest) const {
  const std::vector<Linear*>& ch = lowest->chain_fwd;
  ffh_chain_layer ls[FFH_CHAIN_MAX_LAYERS];
  fill_chain(ch, ls);
  const Tensor& x = lowest->inputs[0];
  const int rc = api->ffh_mlp_chain_fwd(ctx, (const float*)x.impl->ptr, x.impl->ld, ls, (int)ch.size(), local_rows(lowest->outputs[0], this), stream);
  if (rc == FFH_OK) n_chain_fwd_calls++;
  return rc;
}
// The whole backward of the chain whose top layer is `top`: every member's dW / db, the data gradients between them and (unless it is
// discarded) the chain input's.  A member that completes the embedding output gradients (grad_attach_layer) gets its event recorded
// behind the call: the chain's weight-gradient kernel still reads the buffer the next gather overwrites.
int FFModel::run_chain_bwd(Linear* top) {
  const std::vector<Linear*>& ch = top->chain_bwd;
  ffh_chain_layer ls[FFH_CHAIN_MAX_LAYERS];
  fill_chain(ch, ls);
  Linear* lo = ch.front();
  const Tensor& x = lo->inputs[0];
  const int flags = (top->dy_premasked ? FFH_LINEAR_DY_PREMASKED : 0) | (lo->dx_overwrite ? FFH_LINEAR_DX_OVERWRITE : 0) |
                    (lo->dx_mask_by_x ? FFH_LINEAR_DX_MASK_BY_X : 0);
  const int rc = api->ffh_mlp_chain_bwd(ctx, (const float*)x.impl->ptr, x.impl->ld, lo->discard_input_grad ? nullptr : x.impl->grad, x.impl->grad_ld, ls,
                                        (int)ch.size(), local_rows(top->outputs[0], this), flags, stream);
  if (rc == FFH_OK) { n_chain_bwd_calls++; for (Linear* li : ch) li->db_from_upper = false; }
  return rc;
}

// =============================================================================================
// Embedding [ref: src/ops/embedding.cu]
// =============================================================================================
Embedding::Embedding(FFModel& model, const Tensor& input, int _num_entries, int outDim, AggrMode _aggr, const Op* shared_op,
                     Initializer* ki, const char* name)
    : Op(model, OP_EMBEDDING, name, 1, &input), num_entries(_num_entries), out_channels(outDim), aggr(_aggr),
      kernel_initializer(ki) {
  if (shared_op) die("%s: weight sharing is not supported on this path", this->name);
  if (input.data_type != DT_INT64) die("%s: input must be DT_INT64", this->name);
  if (input.numDim != 2) die("%s: input must be [batch][bag]", this->name);
  if (input.owner_op != nullptr) die("%s: input must be a model input tensor", this->name);
  if (aggr != AGGR_MODE_SUM && aggr != AGGR_MODE_AVG) die("%s: aggr must be SUM or AVG", this->name);
  outputs[0].numDim = 2;
  outputs[0].adim[0] = outDim;
  outputs[0].adim[1] = input.adim[1];
  numWeights = 1;
  table_index = (int)model.embeddings.size();
  owner_rank = table_index % model.world_size;   // table i -> GPU i % n [ref: examples/cpp/DLRM/strategies/dlrm_strategy.cc:252-256]
  // giant tables: column-wise instead (the reference cannot split an embedding except on the sample
  // dim, [ref: src/ops/embedding.cu:84-85]); every rank then holds all rows x out_dim/G columns
  column_sharded = model.world_size > 1 && model.config.column_shard_rows > 0 && num_entries >= model.config.column_shard_rows;
  if (column_sharded && outDim % model.world_size != 0) die("%s: out_dim %d is not divisible by %d ranks", this->name, outDim, model.world_size);
  local_cols = column_sharded ? outDim / model.world_size : outDim;
  // ... or row-wise: every rank holds num_entries / G rows of all columns and contributes partial bag sums that a
  // reduce-scatter adds up (BASELINE configs[4]'s "reduce-scatter stress"; column-wise stays the primary layout)
  local_idx = nullptr; partial = gfull = nullptr;
  row_sharded = false; row_begin = 0; rows_local = num_entries;
  set_row_sharding(model, model.exchange && model.config.row_shard_rows > 0 && num_entries >= model.config.row_shard_rows);
  replicated = false;
  set_replicated(model, model.exchange && model.world_size > 1 && !row_sharded && !column_sharded && model.config.replicate_embedding_rows > 0 &&
                            num_entries <= model.config.replicate_embedding_rows);
  model.embeddings.push_back(this);
}
void Embedding::set_replicated(const FFModel& model, bool on) {
  if (on && !model.config.comm.allreduce_sum_f32) die("%s: a data-parallel table needs the allreduce callback of ffcomm", this->name);
  replicated = on;
  if (on) {
    set_row_sharding(model, false);
    column_sharded = false;
    local_cols = out_channels;
    owner_rank = -1;                                   // every rank holds it
  } else if (owner_rank < 0 && !row_sharded) {
    owner_rank = table_index % model.world_size;
  }
}
void Embedding::set_row_sharding(const FFModel& model, bool on) {
  if (on && (!model.config.comm.reduce_scatter_sum_f32 || !model.config.comm.allgather_f32))
    die("%s: --row-shard-rows needs the reduce_scatter / allgather callbacks of ffcomm", this->name);
  if (on && num_entries < model.world_size) die("%s: %d rows cannot be split over %d ranks", this->name, num_entries, model.world_size);
  row_sharded = on;
  if (on) {
    column_sharded = false;
    local_cols = out_channels;
    owner_rank = -1;                                   // no single owner
    row_begin = (int64_t)num_entries * model.rank / model.world_size;
    rows_local = (int64_t)num_entries * (model.rank + 1) / model.world_size - row_begin;
  } else {
    row_begin = 0;
    rows_local = num_entries;
    if (owner_rank < 0) owner_rank = table_index % model.world_size;
  }
}
void Embedding::create_output_and_partition(FFModel&) {}
void Embedding::create_weights(FFModel& model) {
  const int dims[2] = {row_sharded ? (int)rows_local : num_entries, local_cols};   // column- / row-sharded: this rank's slice only
  weights[0] = model.create_weight<2>(dims, this, DT_FLOAT, kernel_initializer);
}
void Embedding::forward(const FFModel& ff) {
  // The first table launches the whole group.  With overlap it goes to the side stream, ordered behind the
  // fork event recorded at the top of forward() (inputs ready): the host has ALREADY enqueued the layers
  // in front of the tables (the bottom MLP), so the GPU runs those while the host walks the exchange callback.
  if (ff.emb_forward_issued) return;
  if (ff.config.overlap_embedding) {
    ff.issue_embedding_forward_on_side_stream();
  } else {
    ff.embedding_group_forward(ff.stream);
    ff.emb_forward_issued = true;
    ff.emb_forward_joined = true;
  }
}
void Embedding::backward(const FFModel& ff) {
  // reverse layer order: the LAST table is visited first; every table's output gradient is
  // complete by then (their only consumer ran already), so the group update can start.
  if (table_index != (int)ff.embeddings.size() - 1) return;
  // data-parallel (replicated) tables: their dense gradient [ref: embed_backward, src/ops/embedding.cu:192-217,308-320] goes
  // into the dense slab on the compute stream; the slab's all-reduce and optimizer launch in update() then treat it like any
  // MLP parameter
  ff.replicated_embedding_grads();
  if (ff.fused_embedding_update()) {
    if (ff.config.overlap_embedding) {
      // gradients of every table are complete here; the side-stream update itself is issued at the END of
      // backward(), after the host has enqueued the bottom-MLP backward it overlaps with
      if (!ff.grad_ready_attached) ff.check(ff.api->ffh_event_record(ff.ctx, ff.ev_grad_ready, ff.stream), "event");
      if (ff.exchange && !ff.config.comm.nonblocking && !ff.use_workers()) {
        ff.emb_update_pending = true;      // host-side collectives on this thread: issue after the bottom-MLP backward is enqueued
      } else {
        ff.issue_embedding_update_on_side_stream();
      }
    }
    return;
  }
  // reference path: dense scatter-add into the full-table gradient [ref: src/ops/embedding.cu:308-320]
  if (ff.exchange) return;      // multi-rank: the rows' gradients first go back to the owners -- embedding_dense_update(), from update()
  for (Embedding* e : ff.embeddings) {
    if (e->owner_rank != ff.rank || e->replicated) continue;
    const Tensor& in = e->inputs[0];
    const Tensor& out = e->outputs[0];
    const float* g = out.impl->grad;
    int64_t gld = out.impl->grad_ld, batch = ff.local_batch;
    ff.check(ff.api->ffh_embedding_bwd_dense(ff.ctx, (const int64_t*)in.impl->ptr, g, e->weights[0].impl->grad, in.adim[0],
                                             e->out_channels, batch, e->num_entries, gld, (int)e->aggr, ff.stream), e->name);
  }
}

// =============================================================================================
// Concat [ref: src/ops/concat.cu]
// =============================================================================================
Concat::Concat(FFModel& model, int n, const Tensor* _inputs, int _axis, const char* name)
    : Op(model, OP_CONCAT, name, n, _inputs) {
  if (n < 1) die("%s: needs at least one input", this->name);
  const int nd = _inputs[0].numDim;
  axis = nd - 1 - _axis;   // user axis -> Legion axis [ref: src/ops/concat.cu:29-49]
  bwd_overwrite = false;
  bwd_done = false;
  if (axis < 0 || axis >= nd) die("%s: axis out of range", this->name);
  outputs[0].numDim = nd;
  for (int d = 0; d < nd; d++) outputs[0].adim[d] = _inputs[0].adim[d];
  for (int i = 1; i < n; i++) {
    if (_inputs[i].numDim != nd) die("%s: rank mismatch", this->name);
    for (int d = 0; d < nd; d++) {
      if (d == axis) outputs[0].adim[d] += _inputs[i].adim[d];
      else if (_inputs[i].adim[d] != outputs[0].adim[d]) die("%s: shape mismatch on dim %d", this->name, d);
    }
  }
  for (int i = 0; i < n; i++)
    if (_inputs[i].data_type != DT_FLOAT) die("%s: inputs must be DT_FLOAT", this->name);
}
void Concat::create_output_and_partition(FFModel&) {}

namespace {
// calc_blk_size [ref: src/ops/concat.cu:194-208]: block = dims <= axis, num_blocks = dims > axis
void concat_geometry(const Concat* c, const FFModel& ff, int64_t& num_blocks, int64_t& out_blk, std::vector<int64_t>& in_blk) {
  const Tensor& o = c->outputs[0];
  num_blocks = 1; out_blk = 1;
  for (int d = 0; d < o.numDim; d++) {
    if (d <= c->axis) out_blk *= o.adim[d];
    else num_blocks *= (d == o.numDim - 1) ? o.adim[d] / ff.world_size : o.adim[d];   // batch is sharded over ranks
  }
  in_blk.resize(c->numInputs);
  for (int i = 0; i < c->numInputs; i++) {
    int64_t b = 1;
    for (int d = 0; d <= c->axis; d++) b *= c->inputs[i].adim[d];
    in_blk[i] = b;
  }
}
}  // namespace

namespace {
// flat part list of a concat: an input scattered over several buffers (column-sharded table) contributes its pieces
void concat_parts(const Concat* c, const std::vector<int64_t>& ib, bool grads, std::vector<float*>& ptrs,
                  std::vector<int64_t>& blks, std::vector<int64_t>& lds) {
  for (int i = 0; i < c->numInputs; i++) {
    const TensorImpl* im = c->inputs[i].impl;
    if (!im->pieces.empty()) {
      if (c->axis != 0) die("%s: a column-sharded input needs a feature-axis concat", c->name);
      for (const TensorPiece& p : im->pieces) { ptrs.push_back(grads ? p.grad : p.ptr); blks.push_back(p.cols); lds.push_back(p.ld); }
    } else {
      ptrs.push_back(grads ? im->grad : (float*)im->ptr);       // NULL gradient (model input): skipped by the kernel
      blks.push_back(ib[i]);
      lds.push_back(c->axis == 0 ? (grads ? im->grad_ld : im->ld) : ib[i]);
    }
  }
}
}  // namespace

void Concat::forward(const FFModel& ff) {
  if (ff.emb_forward_issued && !ff.emb_forward_joined) ff.join_embedding_forward();   // before the first consumer
  // split mode: the slices that layers without an image of their own wrote in place (the bottom MLP's last layer, on the fp32 kernels) get
  // theirs here, whatever launch produced them (allocate(), step 7)
  for (int i : image_inputs) {
    const TensorImpl* im = inputs[i].impl;
    ff.check(ff.api->ffh_convert_f32_to_bf16x3(ff.ctx, (const float*)im->ptr, local_rows(outputs[0], &ff), inputs[i].adim[0], im->ld, ff.stream), name);
  }
  int64_t nb, ob;
  std::vector<int64_t> ib, blks, lds;
  std::vector<float*> ptrs;
  concat_geometry(this, ff, nb, ob, ib);
  concat_parts(this, ib, false, ptrs, blks, lds);
  ff.check(ff.api->ffh_concat_fwd(ff.ctx, (float*)outputs[0].impl->ptr, ob, (const float* const*)ptrs.data(), blks.data(), lds.data(),
                                  (int)ptrs.size(), nb, ff.stream), name);
}
void Concat::backward(const FFModel& ff) {
  if (bwd_done) { bwd_done = false; return; }            // the layer above stored its data gradient into the inputs' buffers itself
  int64_t nb, ob;
  std::vector<int64_t> ib, blks, lds;
  std::vector<float*> ptrs;
  concat_geometry(this, ff, nb, ob, ib);
  concat_parts(this, ib, true, ptrs, blks, lds);
  ff.check(ff.api->ffh_concat_bwd_ex(ff.ctx, outputs[0].impl->grad, ob, ptrs.data(), blks.data(), lds.data(), (int)ptrs.size(), nb,
                                     bwd_overwrite ? FFH_CONCAT_BWD_OVERWRITE : 0, ff.stream), name);
}

// =============================================================================================
// BatchMatmul [ref: src/ops/batch_matmul.cu]
// =============================================================================================
BatchMatmul::BatchMatmul(FFModel& model, const Tensor& A, const Tensor& B, int asd, int bsd)
    : Op(model, OP_BATCHMATMUL, nullptr, 2, std::vector<Tensor>{A, B}.data()), a_seq_length_dim(asd), b_seq_length_dim(bsd) {
  // A (batch, n, k)  B (batch, k, m)  O (batch, n, m) [ref: src/ops/batch_matmul.cu:31-60]
  if (A.numDim != B.numDim || A.numDim < 3) die("%s: operands must both be [batch..][rows][cols]", name);
  if (A.adim[0] != B.adim[1]) die("%s: inner dimensions differ (%d vs %d)", name, A.adim[0], B.adim[1]);
  for (int d = 2; d < A.numDim; d++)
    if (A.adim[d] != B.adim[d]) die("%s: batch dimensions differ", name);
  outputs[0].numDim = A.numDim;
  for (int d = 0; d < A.numDim; d++) outputs[0].adim[d] = A.adim[d];
  outputs[0].adim[0] = B.adim[0];
}
void BatchMatmul::create_output_and_partition(FFModel&) {}
void BatchMatmul::forward(const FFModel& ff) {
  const Tensor &a = inputs[0], &b = inputs[1], &o = outputs[0];
  const int m = b.adim[0], n = a.adim[1], k = a.adim[0];
  const int64_t batch = local_rows(o, &ff) / n;
  ff.check(ff.api->ffh_bmm_fwd(ff.ctx, (float*)o.impl->ptr, (const float*)a.impl->ptr, (const float*)b.impl->ptr, m, n, k, batch,
                               a_seq_length_dim, b_seq_length_dim, ff.seq_length, ff.stream), name);
}
void BatchMatmul::backward(const FFModel& ff) {
  const Tensor &a = inputs[0], &b = inputs[1], &o = outputs[0];
  const int m = b.adim[0], n = a.adim[1], k = a.adim[0];
  const int64_t batch = local_rows(o, &ff) / n;
  if (ff.seq_length >= 0) die("%s: backward does not support seq_length [ref: src/ops/batch_matmul.cu:483-484]", name);
  if (!a.impl->grad || !b.impl->grad) die("%s: backward needs gradients for both operands", name);
  ff.check(ff.api->ffh_bmm_bwd(ff.ctx, o.impl->grad, (const float*)a.impl->ptr, a.impl->grad, (const float*)b.impl->ptr,
                               b.impl->grad, m, n, k, batch, ff.stream), name);
}

// =============================================================================================
// Transpose / Reshape / Flat [ref: src/ops/transpose.cu, reshape.cu, flat.cu]
// =============================================================================================
Transpose::Transpose(FFModel& model, const Tensor& input, const std::vector<int>& _perm, const char* name)
    : Op(model, OP_TRANSPOSE, name, 1, &input) {
  const int nd = input.numDim;
  if ((int)_perm.size() != nd) die("%s: perm has %zu entries for a %d-D tensor", this->name, _perm.size(), nd);
  if (input.data_type != DT_FLOAT) die("%s: input must be DT_FLOAT", this->name);
  if (_perm[0] != 0) die("%s: the batch dimension must stay outermost (it is sharded over ranks)", this->name);
  outputs[0].numDim = nd;
  for (int i = 0; i < nd; i++) {
    perm[i] = _perm[i];
    if (perm[i] < 0 || perm[i] >= nd) die("%s: bad perm", this->name);
    outputs[0].adim[nd - 1 - i] = input.adim[nd - 1 - perm[i]];   // natural dim i of the output = natural dim perm[i] of the input
  }
}
namespace {
void natural_local_dims(const Tensor& t, const FFModel& ff, int64_t* d) {
  for (int i = 0; i < t.numDim; i++) d[i] = t.adim[t.numDim - 1 - i];
  d[0] = d[0] / ff.world_size;   // batch-sharded
}
bool contiguous(const TensorImpl* im, const Tensor& t) { return im->pieces.empty() && im->ld == t.adim[0]; }
}  // namespace
void Transpose::forward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!contiguous(x.impl, x) || !contiguous(y.impl, y)) die("%s: operands must be contiguous", name);
  int64_t d[MAX_TENSOR_DIM];
  natural_local_dims(x, ff, d);
  ff.check(ff.api->ffh_transpose_fwd(ff.ctx, (float*)y.impl->ptr, (const float*)x.impl->ptr, x.numDim, d, perm, ff.stream), name);
}
void Transpose::backward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!x.impl->grad) return;
  if (x.impl->grad_ld != x.adim[0] || y.impl->grad_ld != y.adim[0]) die("%s: gradients must be contiguous", name);
  int64_t d[MAX_TENSOR_DIM];
  natural_local_dims(x, ff, d);
  ff.check(ff.api->ffh_transpose_bwd(ff.ctx, x.impl->grad, y.impl->grad, x.numDim, d, perm, ff.stream), name);
}

Reshape::Reshape(FFModel& model, OperatorType type, const Tensor& input, const std::vector<int>& shape, const char* name)
    : Op(model, type, name, 1, &input) {
  if (input.data_type != DT_FLOAT) die("%s: input must be DT_FLOAT", this->name);
  size_t vol = 1;
  for (int v : shape) vol *= (size_t)v;
  if (vol != input.get_volume()) die("%s: %zu elements cannot be viewed as %zu", this->name, input.get_volume(), vol);
  if (shape.empty() || (int)shape.size() > MAX_TENSOR_DIM) die("%s: 1..%d dimensions", this->name, MAX_TENSOR_DIM);
  if (shape[0] != input.adim[input.numDim - 1]) die("%s: the batch dimension must be preserved (it is sharded over ranks)", this->name);
  outputs[0].numDim = (int)shape.size();
  for (size_t i = 0; i < shape.size(); i++) outputs[0].adim[shape.size() - 1 - i] = shape[i];
  is_view = false;
}
void Reshape::forward(const FFModel& ff) {
  if (is_view) return;
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!contiguous(y.impl, y)) die("%s: output must be contiguous", name);
  const int64_t rows = x.impl->rows_local, cols = x.adim[0];
  if (contiguous(x.impl, x)) {   // copy_kernel [ref: src/ops/reshape.cu:203-210, src/ops/flat.cu:117-124]
    ff.check(ff.api->ffh_memcpy_d2d(ff.ctx, y.impl->ptr, x.impl->ptr, (size_t)rows * cols * sizeof(float), ff.stream), name);
  } else {                       // input lives as a column slice / pieces of other buffers: gather it with the concat kernel
    std::vector<const float*> ptrs; std::vector<int64_t> blks, lds;
    if (!x.impl->pieces.empty()) for (const TensorPiece& p : x.impl->pieces) { ptrs.push_back(p.ptr); blks.push_back(p.cols); lds.push_back(p.ld); }
    else { ptrs.push_back((const float*)x.impl->ptr); blks.push_back(cols); lds.push_back(x.impl->ld); }
    ff.check(ff.api->ffh_concat_fwd(ff.ctx, (float*)y.impl->ptr, cols, ptrs.data(), blks.data(), lds.data(), (int)ptrs.size(), rows, ff.stream), name);
  }
}
void Reshape::backward(const FFModel& ff) {
  if (is_view) return;          // the gradient of the view is the gradient of the tensor
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!x.impl->grad && x.impl->pieces.empty()) return;
  const int64_t rows = x.impl->rows_local, cols = x.adim[0];
  if (x.impl->pieces.empty() && x.impl->grad_ld == cols) {   // in_grad += out_grad
    ff.check(ff.api->ffh_add_scaled(ff.ctx, x.impl->grad, y.impl->grad, rows * cols, 1.0f, ff.stream), name);
  } else {
    std::vector<float*> ptrs; std::vector<int64_t> blks, lds;
    if (!x.impl->pieces.empty()) for (const TensorPiece& p : x.impl->pieces) { ptrs.push_back(p.grad); blks.push_back(p.cols); lds.push_back(p.ld); }
    else { ptrs.push_back(x.impl->grad); blks.push_back(cols); lds.push_back(x.impl->grad_ld); }
    ff.check(ff.api->ffh_concat_bwd(ff.ctx, y.impl->grad, cols, ptrs.data(), blks.data(), lds.data(), (int)ptrs.size(), rows, ff.stream), name);
  }
}

Tril::Tril(FFModel& model, const Tensor& input, const char* name) : Op(model, OP_TRIL, name, 1, &input) {
  if (input.data_type != DT_FLOAT) die("%s: input must be DT_FLOAT", this->name);
  if (input.numDim != 3 || input.adim[0] != input.adim[1]) die("%s: input must be [batch][n][n]", this->name);
  n = input.adim[0];
  if (n < 2 || n > 64) die("%s: n = %d, supported 2..64", this->name, n);
  outputs[0].numDim = 2;
  outputs[0].adim[0] = n * (n - 1) / 2;
  outputs[0].adim[1] = input.adim[2];
}
void Tril::forward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!contiguous(x.impl, x) || !y.impl->pieces.empty()) die("%s: input must be contiguous", name);
  ff.check(ff.api->ffh_tril_fwd(ff.ctx, (float*)y.impl->ptr, y.impl->ld, (const float*)x.impl->ptr, x.impl->rows_local / n, n, ff.stream), name);
}
void Tril::backward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!x.impl->grad) return;
  if (x.impl->grad_ld != x.adim[0]) die("%s: input gradient must be contiguous", name);
  ff.check(ff.api->ffh_tril_bwd(ff.ctx, x.impl->grad, y.impl->grad, y.impl->grad_ld, x.impl->rows_local / n, n, ff.stream), name);
}
Tensor FFModel::tril(const Tensor& input, const char* name) {
  Tril* t = new Tril(*this, input, name);
  t->layer_index = (int)layers.size();
  layers.push_back(t);
  return t->outputs[0];
}
DotInteraction::DotInteraction(FFModel& model, const Tensor& input, int _d, const char* name)
    : Op(model, OP_DOT_INTERACTION, name, 1, &input), d(_d), bwd_overwrite(false) {
  if (input.data_type != DT_FLOAT) die("%s: input must be DT_FLOAT", this->name);
  if (input.numDim != 2 || d < 1 || input.adim[0] % d != 0) die("%s: input must be [batch][c * %d]", this->name, d);
  c = input.adim[0] / d;
  if (c < 2 || c > 32) die("%s: %d feature vectors, supported 2..32", this->name, c);
  outputs[0].numDim = 2;
  outputs[0].adim[0] = d + c * (c - 1) / 2;
  outputs[0].adim[1] = input.adim[1];
}
void DotInteraction::forward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!x.impl->pieces.empty() || !y.impl->pieces.empty()) die("%s: operands must be single buffers", name);
  ff.check(ff.api->ffh_dot_interaction_fwd(ff.ctx, (const float*)x.impl->ptr, x.impl->ld, (float*)y.impl->ptr, y.impl->ld, x.impl->rows_local, c, d,
                                           ff.stream), name);
}
void DotInteraction::backward(const FFModel& ff) {
  const Tensor &x = inputs[0], &y = outputs[0];
  if (!x.impl->grad) return;
  ff.check(ff.api->ffh_dot_interaction_bwd(ff.ctx, (const float*)x.impl->ptr, x.impl->ld, y.impl->grad, y.impl->grad_ld, x.impl->grad, x.impl->grad_ld,
                                           x.impl->rows_local, c, d, bwd_overwrite ? FFH_DOT_BWD_OVERWRITE : 0, ff.stream), name);
}
Tensor FFModel::dot_interaction(const Tensor& input, int d, const char* name) {
  DotInteraction* t = new DotInteraction(*this, input, d, name);
  t->layer_index = (int)layers.size();
  layers.push_back(t);
  return t->outputs[0];
}
Tensor FFModel::transpose(const Tensor& input, const std::vector<int>& perm, const char* name) {
  Transpose* t = new Transpose(*this, input, perm, name);
  t->layer_index = (int)layers.size();
  layers.push_back(t);
  return t->outputs[0];
}
Tensor FFModel::reshape(const Tensor& input, const std::vector<int>& shape, const char* name) {
  Reshape* r = new Reshape(*this, OP_RESHAPE, input, shape, name);
  r->layer_index = (int)layers.size();
  layers.push_back(r);
  return r->outputs[0];
}
Tensor FFModel::flat(const Tensor& input, const char* name) {
  // [batch][rest...] -> [batch][prod(rest)] [ref: src/ops/flat.cu:31-60]
  const int batch = input.adim[input.numDim - 1];
  Reshape* r = new Reshape(*this, OP_FLAT, input, {batch, (int)(input.get_volume() / (size_t)batch)}, name);
  r->layer_index = (int)layers.size();
  layers.push_back(r);
  return r->outputs[0];
}

// Parameters that live in the dense slab (one all-reduce bucket, one optimizer launch): Linear weights / biases and the
// tables of data-parallel (replicated) embeddings.
static bool in_dense_slab(const Parameter& p) {
  if (p.owner_op->op_type == OP_LINEAR) return true;
  return p.owner_op->op_type == OP_EMBEDDING && static_cast<const Embedding*>(p.owner_op)->replicated;
}

// =============================================================================================
// SGDOptimizer [ref: src/runtime/optimizer.cc:43-189]
// =============================================================================================
SGDOptimizer::SGDOptimizer(const FFModel* _model, double _lr, double _momentum, bool _nesterov, double _wd)
    : Optimizer(_model), lr(_lr), momentum(_momentum), nesterov(_nesterov), weight_decay(_wd) {}
void SGDOptimizer::init(void) {
  if (momentum > 0.0) {
    for (const Parameter& p : model->parameters) {
      if (!p.impl->grad) continue;
      const size_t bytes = p.impl->bytes;          // (rows x leading dimension: a padded Linear kernel keeps its pad columns)
      float* v = (float*)model->dmalloc(bytes);
      model->check(model->api->ffh_zero(model->ctx, v, bytes, model->stream), "momentum init");
      v_values[p.impl->ptr] = v;
    }
  }
}
void SGDOptimizer::next(void) {}
void SGDOptimizer::update(const Parameter* p) {
  if (!p->impl->grad) return;   // embedding tables on the fused path have no dense gradient
  float* v = momentum > 0.0 ? v_values[p->impl->ptr] : nullptr;
  model->check(model->api->ffh_sgd_update(model->ctx, (float*)p->impl->ptr, p->impl->grad, v, (int64_t)(p->impl->bytes / sizeof(float)), (float)lr,
                                          (float)weight_decay, (float)momentum, nesterov ? 1 : 0, model->stream), "sgd_update");
}

// =============================================================================================
// AdamOptimizer [ref: src/runtime/optimizer.cc:190-330, src/runtime/optimizer_kernel.cu:206-226]
// =============================================================================================
AdamOptimizer::AdamOptimizer(const FFModel* _model, double _alpha, double _beta1, double _beta2, double _wd, double _eps)
    : Optimizer(_model), alpha(_alpha), beta1(_beta1), beta2(_beta2), weight_decay(_wd), epsilon(_eps), alpha_t(_alpha),
      beta1_t(1.0f), beta2_t(1.0f), mlp_m(nullptr), mlp_v(nullptr) {}
void AdamOptimizer::init(void) {
  // ZeroInitializer on both moments of every parameter [ref: optimizer.cc:204-236]
  auto zeros = [&](size_t count) {
    float* p = (float*)model->dmalloc(count * sizeof(float));
    model->check(model->api->ffh_zero(model->ctx, p, count * sizeof(float), model->stream), "adam moments");
    return p;
  };
  if (model->mlp_count) { mlp_m = zeros(model->mlp_count); mlp_v = zeros(model->mlp_count); }
  for (const Parameter& p : model->parameters) {
    if (in_dense_slab(p) || !p.impl->grad) continue;
    mv_values[p.impl->ptr] = std::make_pair(zeros(p.impl->bytes / sizeof(float)), zeros(p.impl->bytes / sizeof(float)));
  }
}
void AdamOptimizer::next(void) {
  // [ref: optimizer.cc:248-254]
  beta1_t *= beta1;
  beta2_t *= beta2;
  alpha_t = alpha * sqrt(1 - beta2_t) / (1 - beta1_t);
}
void AdamOptimizer::update(const Parameter* p) {
  if (!p->impl->grad) return;
  float *m, *v;
  if (in_dense_slab(*p)) {
    const size_t off = (size_t)((float*)p->impl->ptr - model->mlp_weights);
    m = mlp_m + off; v = mlp_v + off;
  } else {
    auto it = mv_values.find(p->impl->ptr);
    if (it == mv_values.end()) die("AdamOptimizer::update: parameter without moments");
    m = it->second.first; v = it->second.second;
  }
  model->check(model->api->ffh_adam_update(model->ctx, (float*)p->impl->ptr, p->impl->grad, m, v, (int64_t)(p->impl->bytes / sizeof(float)), (float)alpha_t,
                                           (float)beta1, (float)beta2, (float)weight_decay, (float)epsilon, 0, model->stream), "adam_update");
}

// =============================================================================================
// compile / allocate
// =============================================================================================
int FFModel::tables_of_rank(int r) const {
  int n = 0;
  for (const Embedding* e : embeddings) n += e->owner_rank == r;
  return n;
}

int FFModel::next_seed() {
  return (int)(ffh_hash(config.seed * 0x9E3779B97F4A7C15ULL + 0x5EED, seed_counter++) & 0x7fffffffULL);
}

bool FFModel::fused_embedding_update() const {
  if (config.dense_embedding_update) return false;
  const SGDOptimizer* sgd = dynamic_cast<const SGDOptimizer*>(optimizer);
  // the fused sparse update equals the reference's dense sweep only for plain SGD ...
  if (sgd && sgd->momentum == 0.0 && sgd->weight_decay == 0.0) return true;
  // ... every other optimizer takes the reference's dense path (zero + scatter-add + whole-table sweep: reference semantics on
  // every row) unless the user opts into the touched-rows rule (--sparse-embedding-optimizer; stated divergence: ffh_sparse_opt)
  return config.sparse_embedding_optimizer && (sgd || dynamic_cast<const AdamOptimizer*>(optimizer));
}

// the row rule of the sorted-segments update for the optimizer in force; false: plain SGD (the lr-only entry points)
bool FFModel::sparse_rule(ffh_sparse_opt& o) const {
  memset(&o, 0, sizeof o);
  if (const SGDOptimizer* sgd = dynamic_cast<const SGDOptimizer*>(optimizer)) {
    o.lr = (float)sgd->lr;
    if (sgd->momentum == 0.0 && sgd->weight_decay == 0.0) { o.kind = FFH_SPARSE_OPT_SGD; return false; }
    o.kind = FFH_SPARSE_OPT_SGD_MOMENTUM; o.weight_decay = (float)sgd->weight_decay; o.momentum = (float)sgd->momentum; o.nesterov = sgd->nesterov ? 1 : 0;
    return true;
  }
  const AdamOptimizer* adam = dynamic_cast<const AdamOptimizer*>(optimizer);
  if (!adam) die("sparse_rule: unknown optimizer");
  // alpha_t of THIS step [ref: AdamOptimizer::next, src/runtime/optimizer.cc:248-254].  The reference advances it at the top of
  // update(); the side-stream table update is issued from backward(), before that -- it looks one next() ahead then.
  double alpha_t = adam->alpha_t;
  if (!opt_next_done) {
    const double b1 = adam->beta1_t * adam->beta1, b2 = adam->beta2_t * adam->beta2;
    alpha_t = adam->alpha * sqrt(1 - b2) / (1 - b1);
  }
  o.kind = FFH_SPARSE_OPT_ADAM; o.lr = (float)alpha_t; o.weight_decay = (float)adam->weight_decay;
  o.beta1 = (float)adam->beta1; o.beta2 = (float)adam->beta2; o.epsilon = (float)adam->epsilon;
  return true;
}

// Placement from a strategy file [ref: FFModel::compile -> load_strategies_from_file, src/runtime/model.cc:1575-1577;
// Op::create_output_and_partition looks its config up by op name, e.g. src/ops/embedding.cu:75-79].  Ops the file
// does not name keep the default: tables round-robin over the ranks, everything else data-parallel.
void FFModel::apply_strategies() {
  if (!config.import_strategy_file.empty() && !load_strategies_from_file(config.import_strategy_file, config.strategies))
    die("cannot read strategy file %s", config.import_strategy_file.c_str());
  for (Op* op : layers) {
    ParallelConfig pc;
    if (!config.find_parallel_config(op->outputs[0].numDim, op->name, pc)) continue;
    if (pc.device_type != ParallelConfig::GPU) die("%s: strategy places it on the CPU; this build runs every op on the GPUs", op->name);
    for (int id : pc.device_ids)
      if (id < 0 || id >= world_size) die("%s: strategy names device %d, the job has %d rank(s)", op->name, id, world_size);
    if (Embedding* e = dynamic_cast<Embedding*>(op)) {
      // one table on one device (what dlrm_strategy.cc emits), or split over the sample dim: a data-parallel table, replicated
      // with an all-reduced dense gradient -- what the reference does with an op that has no strategy entry
      if (pc.num_parts() == world_size && world_size > 1 && pc.is_data_parallel()) {
        for (size_t j = 0; j < pc.device_ids.size(); j++)
          if (pc.device_ids[j] != (int)j) die("%s: data-parallel parts must sit on devices 0..%d in order", op->name, world_size - 1);
        e->set_replicated(*this, true);
        continue;
      }
      if (pc.num_parts() == world_size && world_size > 1 && pc.dim[0] == world_size) {
        // this build's own extension, as --export writes it: the table split column-wise over all ranks
        if (e->out_channels % world_size) die("%s: out_dim %d is not divisible by %d ranks", op->name, e->out_channels, world_size);
        for (size_t j = 0; j < pc.device_ids.size(); j++)
          if (pc.device_ids[j] != (int)j) die("%s: column blocks must sit on devices 0..%d in order", op->name, world_size - 1);
        e->set_replicated(*this, false);
        e->set_row_sharding(*this, false);
        e->column_sharded = true;
        e->local_cols = e->out_channels / world_size;
        continue;
      }
      if (pc.num_parts() != 1) die("%s: an embedding can only be placed whole on one device (dims all 1), the strategy splits it %d ways", op->name, pc.num_parts());
      e->set_replicated(*this, false);
      e->set_row_sharding(*this, false);
      e->owner_rank = pc.device_ids.empty() ? 0 : pc.device_ids[0];
      e->column_sharded = false;
      e->local_cols = e->out_channels;
    } else {
      if (!pc.is_data_parallel() || pc.num_parts() != world_size)
        die("%s: only data parallelism over all %d rank(s) is built for this op (strategy: %d parts%s)", op->name, world_size, pc.num_parts(),
            pc.is_data_parallel() ? "" : ", not on the sample dim");
      for (size_t j = 0; j < pc.device_ids.size(); j++)
        if (pc.device_ids[j] != (int)j) die("%s: data-parallel parts must sit on devices 0..%d in order", op->name, world_size - 1);
    }
  }
  if (!config.export_strategy_file.empty() && rank == 0) {
    // the placement in force, in the reference's format (it writes the result of its search here)
    std::map<std::string, ParallelConfig> out;
    for (Op* op : layers) {
      ParallelConfig pc;
      pc.nDims = op->outputs[0].numDim;
      Embedding* e = dynamic_cast<Embedding*>(op);
      if (e && e->row_sharded) continue;              // no output dim is split: the file format cannot say it; the flag stays in charge
      if (e && !e->column_sharded && !e->replicated) {
        pc.device_ids.push_back(e->owner_rank);
      } else if (e && e->column_sharded) {
        pc.dim[0] = world_size;                       // column-wise giant table: split on the channel dim (this build's extension)
        for (int j = 0; j < world_size; j++) pc.device_ids.push_back(j);
      } else {
        pc.dim[pc.nDims - 1] = world_size;
        for (int j = 0; j < world_size; j++) pc.device_ids.push_back(j);
      }
      out[op->name] = pc;
    }
    if (!save_strategies_to_file(config.export_strategy_file, out)) die("cannot write strategy file %s", config.export_strategy_file.c_str());
  }
}

void FFModel::compile(LossType lt, const std::vector<MetricsType>& metrics, CompMode cm) {
  if (!optimizer) die("compile(): no optimizer set");
  compile(optimizer, lt, metrics, cm);
}

void FFModel::compile(Optimizer* _optimizer, LossType _loss_type, const std::vector<MetricsType>& metrics, CompMode comp_mode) {
  if (compiled) die("compile() called twice");
  if (layers.empty()) die("compile(): the model has no layers");
  apply_strategies();
  optimizer = _optimizer;
  loss_type = _loss_type;
  config.computationMode = comp_mode;
  if (loss_type != LOSS_MEAN_SQUARED_ERROR_AVG_REDUCE && loss_type != LOSS_MEAN_SQUARED_ERROR_SUM_REDUCE)
    die("loss type %d is not on the DLRM path (only the two MSE losses)", (int)loss_type);
  metrics_flags = 0;
  for (MetricsType m : metrics) {
    switch (m) {
      case METRICS_ACCURACY: metrics_flags |= 1; break;
      case METRICS_MEAN_SQUARED_ERROR: metrics_flags |= 2; break;
      case METRICS_ROOT_MEAN_SQUARED_ERROR: metrics_flags |= 4; break;
      case METRICS_MEAN_ABSOLUTE_ERROR: metrics_flags |= 8; break;
      default: die("metrics type %d is not on the DLRM path", (int)m);
    }
  }
  for (Op* op : layers) {
    op->create_output_and_partition(*this);
    op->create_weights(*this);
    for (int i = 0; i < op->numWeights; i++) parameters.push_back(op->weights[i]);
  }
  // label tensor: same shape as the final output [ref: src/runtime/model.cc:1740-1769]
  {
    const Tensor& fin = layers.back()->outputs[0];
    label_tensor = fin;
    label_tensor.owner_op = nullptr;
    label_tensor.impl = new TensorImpl();
    tensor_impls.push_back(label_tensor.impl);
    label_tensor.impl->is_input = true;
  }
  // Collectives served by host callbacks (torch.distributed) cannot be captured.  RcclComm's are plain enqueues on the model's own
  // streams from C++ (ffcomm.nonblocking): with --capture-exchange the per-rank step -- kernels, both all-to-alls, the all-reduce --
  // is captured and replayed as one hipGraph, as the reference wraps every iteration in a Legion trace on any GPU count
  // [ref: examples/cpp/DLRM/dlrm.cc:174-181].  Behind a flag until a multi-GPU box has measured it.
  if (exchange && config.enable_graph && !(config.capture_exchange && config.comm.nonblocking)) config.enable_graph = false;
  // Measured on ROCm 7.0 (the runtime torch bundles): hipStreamEndCapture recurses without end (174,586 frames of
  // hip::Stream::EndCapture, profiles/r04_capture_exchange_endcapture_backtrace.txt) when RCCL's grouped send / recv were captured
  // on a stream that itself joined the capture through an event -- the side stream of the overlapped gather.  With the collectives
  // on the capturing stream itself the capture works, so a captured exchange step runs its embedding branch on the compute stream.
  // (first: Adam never captures -- alpha_t is a new launch argument every step -- so it must not lose the side-stream overlap to a
  //  capture that will not happen; round-4 advisor)
  if (dynamic_cast<AdamOptimizer*>(optimizer) && config.enable_graph) config.enable_graph = false;
  if (exchange && config.enable_graph && config.capture_exchange) config.overlap_embedding = false;
  // Any optimizer x any placement (round 4).  Plain SGD: the fused sorted-segments update.  Momentum / weight-decay SGD, Adam:
  // by default the reference's own path on the rank(s) that hold the table -- an owner-local dense gradient (zeroed, scatter-added
  // from the rows the backward all-to-all returned, swept by sgd_update / adam_update with dense per-table state; sole owner: no
  // all-reduce) -- or, with --sparse-embedding-optimizer, the touched-rows rule on the sorted segments with per-row state
  // (ffh_sparse_opt).  Data-parallel (replicated) tables live in the dense slab and follow the MLP's optimizer launch either way.
  allocate();
  for (Op* op : layers) {
    if (Linear* li = dynamic_cast<Linear*>(op)) {
      if (li->in_padded != li->in_channels) {
        // padded kernel: the initializer fills a contiguous [out][in] temporary exactly as it would fill the reference's tensor; the
        // rows are then copied into the padded storage (whose pad columns stay zero)
        Parameter tmp = li->weights[0];
        TensorImpl ti = *li->weights[0].impl;
        ti.ptr = dmalloc(tmp.get_volume() * sizeof(float)); ti.ld = li->in_channels; ti.grad = nullptr;
        tmp.impl = &ti;
        li->kernel_initializer->init(this, &tmp);
        for (int r = 0; r < li->out_channels; r++)
          check(api->ffh_memcpy_d2d(ctx, (float*)li->weights[0].impl->ptr + (size_t)r * li->in_padded, (const float*)ti.ptr + (size_t)r * li->in_channels,
                                    (size_t)li->in_channels * sizeof(float), stream), "padded kernel init");
        check(api->ffh_stream_sync(ctx, stream), "padded kernel init");
        api->ffh_free(ctx, ti.ptr);
        note_weight_write(li->weights[0].impl->ptr);
      } else {
        li->kernel_initializer->init(this, &li->weights[0]);
      }
      if (li->use_bias) li->bias_initializer->init(this, &li->weights[1]);
    } else if (Embedding* e = dynamic_cast<Embedding*>(op)) {
      if (e->held_here(rank)) e->kernel_initializer->init(this, &e->weights[0]);
    }
  }
  compiled = true;
  optimizer->init();
  {   // per-row optimizer state of the touched-rows rule: the shape of the local table (+ the zero row of a row block)
    ffh_sparse_opt rule;
    if (fused_embedding_update() && sparse_rule(rule)) {
      const int nstate = rule.kind == FFH_SPARSE_OPT_ADAM ? 2 : (rule.momentum > 0.0f ? 1 : 0);
      for (Embedding* e : embeddings) {
        if (!e->held_here(rank) || e->replicated) continue;
        const size_t bytes = e->weights[0].impl->bytes + (e->row_sharded ? (size_t)e->out_channels * 4 : 0);
        for (int k = 0; k < nstate; k++) {
          e->opt_state[k] = (float*)dmalloc(bytes);
          check(api->ffh_zero(ctx, e->opt_state[k], bytes, stream), "sparse optimizer state");
        }
      }
    }
  }
  check(api->ffh_stream_sync(ctx, stream), "compile sync");
}

void FFModel::allocate() {
  // ---- 1. who consumes what -------------------------------------------------------------------
  std::map<TensorImpl*, int> consumers;
  for (Op* op : layers)
    for (int i = 0; i < op->numInputs; i++) consumers[op->inputs[i].impl]++;

  // ---- 2. inputs and label ----------------------------------------------------------------------
  std::map<TensorImpl*, Embedding*> sparse_of;
  for (Embedding* e : embeddings) sparse_of[e->inputs[0].impl] = e;
  auto alloc_rows = [&](const Tensor& t, int64_t nrows) {
    TensorImpl* im = t.impl;
    im->ld = t.adim[0];
    im->rows_local = nrows;
    im->bytes = (size_t)nrows * (size_t)t.adim[0] * dtype_size(t.data_type);
    im->ptr = dmalloc(im->bytes);
    check(api->ffh_zero(ctx, im->ptr, im->bytes, stream), "zero input");
  };
  for (Tensor* t : input_tensors) {
    if (t->adim[t->numDim - 1] != config.batchSize)
      die("input tensor %d: outermost dimension %d is not the batch size %d", t->impl->guid, t->adim[t->numDim - 1], config.batchSize);
    auto it = sparse_of.find(t->impl);
    if (it != sparse_of.end()) {
      // sparse ids of a table: the owner gathers for the GLOBAL batch; other ranks hold nothing
      if (it->second->held_here(rank)) alloc_rows(*t, t->rows());
    } else {
      alloc_rows(*t, t->rows() / world_size);
    }
  }
  alloc_rows(label_tensor, label_tensor.rows() / world_size);

  // ---- 3. exchange buffers (table-wise sharding) ----------------------------------------------
  const int T = (int)embeddings.size();
  int D = 0, L = 0;
  if (T) {
    D = embeddings[0]->out_channels;
    L = embeddings[0]->inputs[0].adim[0];
    for (Embedding* e : embeddings)
      if (e->out_channels != D || e->inputs[0].adim[0] != L || e->aggr != embeddings[0]->aggr)
        die("all embedding tables must share out_dim, bag size and aggregation (DLRM)");
    if (T > FFH_MAX_TABLES * 8) die("too many embedding tables");
  }
  owned_tables.clear();
  for (Embedding* e : embeddings)
    if (e->owner_rank == rank) owned_tables.push_back(e->table_index);
  // exchange units: a table lives whole on one rank (table-wise) or as G column blocks (column-wise)
  shards.clear();
  rank_width.assign(world_size, 0);
  for (Embedding* e : embeddings) {
    if (e->row_sharded) {
      // not part of the all-to-all: its own buffers, a reduce-scatter forward and an all-gather backward
      const size_t ids = (size_t)config.batchSize * L, fl = (size_t)config.batchSize * D;
      e->local_idx = (int64_t*)dmalloc(ids * sizeof(int64_t));
      e->partial = (float*)dmalloc(fl * 4);
      e->gfull = (float*)dmalloc(fl * 4);
    } else if (e->replicated) {
      // data-parallel: every rank gathers its own samples from its copy; nothing of it crosses the all-to-all
    } else if (e->column_sharded) {
      for (int g = 0; g < world_size; g++) shards.push_back({e, g, g * e->local_cols, e->local_cols, 0});
    } else {
      shards.push_back({e, e->owner_rank, 0, e->out_channels, 0});
    }
  }
  for (EmbShard& sh : shards) { sh.off = rank_width[sh.owner]; rank_width[sh.owner] += sh.cols; }
  int owned_shards = 0;
  for (const EmbShard& sh : shards) owned_shards += sh.owner == rank;
  if (exchange && T) {
    const size_t send_floats = (size_t)config.batchSize * rank_width[rank];     // [B_global][width of this rank]
    size_t recv_floats = 0;
    fwd_send_counts.assign(world_size, local_batch * rank_width[rank]);
    fwd_recv_counts.resize(world_size);
    for (int s = 0; s < world_size; s++) { fwd_recv_counts[s] = local_batch * rank_width[s]; recv_floats += fwd_recv_counts[s]; }
    xsend = (float*)dmalloc(std::max<size_t>(send_floats, 1) * 4);
    grecv = (float*)dmalloc(std::max<size_t>(send_floats, 1) * 4);
    xrecv = (float*)dmalloc(std::max<size_t>(recv_floats, 1) * 4);
    gsend = (float*)dmalloc(std::max<size_t>(recv_floats, 1) * 4);
  }

  // ---- 4. activations: aliasing into the concat buffer, then one slab per kind -----------------
  // A producer (Linear / Embedding) whose only consumer is a feature-axis Concat writes straight
  // into the Concat output; its gradient is the matching slice of the Concat output gradient.
  std::map<TensorImpl*, std::pair<Concat*, int64_t>> alias_of;   // impl -> (concat, column offset)
  for (Op* op : layers) {
    Concat* c = dynamic_cast<Concat*>(op);
    if (!c || c->axis != 0) continue;
    int64_t off = 0;
    for (int i = 0; i < c->numInputs; i++) {
      const Tensor& in = c->inputs[i];
      const bool producer_ok = in.owner_op && (in.owner_op->op_type == OP_LINEAR || in.owner_op->op_type == OP_EMBEDDING || in.owner_op->op_type == OP_TRIL);
      // (a row-sharded table's output is the contiguous receive buffer of its reduce-scatter: own storage as well)
      const bool via_exchange = exchange && in.owner_op && in.owner_op->op_type == OP_EMBEDDING && !static_cast<const Embedding*>(in.owner_op)->replicated;
      if (producer_ok && !via_exchange && consumers[in.impl] == 1 && !alias_of.count(in.impl)) alias_of[in.impl] = {c, off};
      off += in.adim[0];
    }
  }
  // ---- 4a. reduction depths the persistent GEMMs cannot take (in % 64 != 0): pad the operand, not the kernel -------------
  // MLPerf-DLRM's first top layer reads the dot interaction's 479 columns: K = 479 is not a multiple of the 64-deep k-tiles and
  // rows of 479 floats start at odd dwords, so all three GEMMs of the layer fell back to the register-staged kernels (97 / 75 /
  // 74 TFLOP/s where hipBLASLt does 118 / 118 / 113).  This layer owns both allocations: when the input tensor has storage of its
  // own, one consumer, and a producer that writes with a leading dimension (the interaction, a Linear), the tensor and its
  // gradient get ld = in rounded up to 64 with zero pad columns, and the kernel is stored [out][in_padded] with zero pads -- to the
  // kernel library it is a 512-wide layer.  Same values: the pads add exact zeros at the END of every k sum (x_pad w_pad = 0);
  // dW's pad columns are dy^T x_pad = 0, so the pads stay zero under SGD / momentum / weight decay / Adam; dX's pad columns are
  // dy w_pad = 0 and nobody reads them.  The reference-visible shape stays [out][in] (get / set_weights copy rows).
  std::map<TensorImpl*, int64_t> padded_ld;
  for (Op* op : layers) {
    Linear* li = dynamic_cast<Linear*>(op);
    if (!li) continue;
    li->in_padded = li->in_channels;
    const Tensor& x = li->inputs[0];
    const int fast_in = api->ffh_linear_fast_in_dim(li->in_channels, li->out_channels);      // the library's own padding rule (fast-path contract, ff_hip.h)
    if (!config.pad_linear_k || fast_in == li->in_channels) continue;
    if (!x.owner_op || consumers[x.impl] != 1 || alias_of.count(x.impl) || !x.impl->pieces.empty()) continue;
    if (x.owner_op->op_type != OP_DOT_INTERACTION && x.owner_op->op_type != OP_LINEAR) continue;
    if (x.numDim != 2) continue;
    li->in_padded = fast_in;
    padded_ld[x.impl] = li->in_padded;
  }
  auto cols_of = [&](const Tensor& o) -> int64_t { auto it = padded_ld.find(o.impl); return it == padded_ld.end() ? (int64_t)o.adim[0] : it->second; };
  // sizes
  size_t act_bytes = 0;
  act_grad_bytes = 0;
  std::vector<Op*> need;   // ops whose output gets its own storage
  // Reshape / Flat of a tensor that owns contiguous storage and is read by nothing else: the output is a VIEW of it
  // (the reference copies, src/ops/reshape.cu:203-210 / flat.cu:117-124: same values, two passes over the tensor less)
  std::vector<Reshape*> views;
  for (Op* op : layers) {
    Reshape* r = dynamic_cast<Reshape*>(op);
    if (!r) continue;
    r->is_view = false;
    const Tensor& x = r->inputs[0];
    if (!x.owner_op || consumers[x.impl] != 1 || alias_of.count(x.impl) || alias_of.count(r->outputs[0].impl)) continue;
    if (exchange && x.owner_op->op_type == OP_EMBEDDING) continue;
    if (x.get_volume() != r->outputs[0].get_volume()) continue;
    if (padded_ld.count(x.impl)) continue;          // padded rows are not one contiguous run
    r->is_view = true;
    views.push_back(r);
  }
  for (Op* op : layers) {
    TensorImpl* im = op->outputs[0].impl;
    if (alias_of.count(im)) continue;
    if (Reshape* r = dynamic_cast<Reshape*>(op)) if (r->is_view) continue;
    if (exchange && op->op_type == OP_EMBEDDING && !static_cast<Embedding*>(op)->row_sharded && !static_cast<Embedding*>(op)->replicated) continue;   // lives in xrecv / gsend
    const Tensor& o = op->outputs[0];
    const size_t b = align_up((size_t)(o.rows() / world_size) * cols_of(o) * 4);
    act_bytes += b;
    act_grad_bytes += b;
    need.push_back(op);
  }
  act_slab = (char*)dmalloc(std::max<size_t>(act_bytes, 256));
  act_grad_slab = (char*)dmalloc(std::max<size_t>(act_grad_bytes, 256));
  size_t off_a = 0;
  for (Op* op : need) {
    const Tensor& o = op->outputs[0];
    TensorImpl* im = o.impl;
    const size_t raw = (size_t)(o.rows() / world_size) * cols_of(o) * 4;
    im->ptr = act_slab + off_a;
    im->ld = cols_of(o);
    im->grad = (float*)(act_grad_slab + off_a);
    im->grad_ld = cols_of(o);
    im->bytes = raw;
    im->rows_local = o.rows() / world_size;
    im->alias = true;        // slab-owned: not freed individually
    off_a += align_up(raw);
  }
  for (Reshape* r : views) {     // layer order: a view of a view resolves to the first owner
    const TensorImpl* xi = r->inputs[0].impl;
    const Tensor& o = r->outputs[0];
    TensorImpl* im = o.impl;
    im->ptr = xi->ptr; im->grad = xi->grad;
    im->ld = im->grad_ld = o.adim[0];
    im->rows_local = o.rows() / world_size;
    im->bytes = xi->bytes;
    im->alias = im->grad_alias = true;
  }
  for (auto& kv : alias_of) {
    TensorImpl* im = kv.first;
    Concat* c = kv.second.first;
    TensorImpl* zo = c->outputs[0].impl;
    im->ptr = (float*)zo->ptr + kv.second.second;
    im->ld = zo->ld;
    im->grad = zo->grad + kv.second.second;
    im->grad_ld = zo->grad_ld;
    im->alias = im->grad_alias = true;
    im->rows_local = c->outputs[0].rows() / world_size;
    im->bytes = (size_t)(c->outputs[0].rows() / world_size) * zo->ld * 4;
  }
  if (exchange) {
    // embedding outputs are views into the receive buffer: the block of source s is [Bl][rank_width[s]]
    std::vector<int64_t> base(world_size, 0);
    for (int s = 1; s < world_size; s++) base[s] = base[s - 1] + fwd_recv_counts[s - 1];
    for (const EmbShard& sh : shards) {
      TensorImpl* im = sh.e->outputs[0].impl;
      float* p = xrecv + base[sh.owner] + sh.off;
      float* gp = gsend + base[sh.owner] + sh.off;
      im->rows_local = local_batch;
      im->alias = im->grad_alias = true;
      if (sh.e->column_sharded) {
        im->pieces.push_back({p, gp, rank_width[sh.owner], sh.cols});   // pushed in column order (owner ascending)
        im->bytes = 0;
      } else {
        im->ptr = p; im->ld = rank_width[sh.owner];
        im->grad = gp; im->grad_ld = im->ld;
        im->bytes = (size_t)local_batch * im->ld * 4;
      }
    }
  }

  // ---- 4b. which activation gradients have exactly one producer (then nothing needs zeroing) ----
  need_zero_act_grads = false;
  need_zero_gsend = false;
  for (Op* op : layers) {
    if (consumers[op->outputs[0].impl] > 1) need_zero_act_grads = true;           // several ops add into its gradient
    if (op->op_type == OP_BATCHMATMUL || op->op_type == OP_TRANSPOSE || op->op_type == OP_RESHAPE || op->op_type == OP_FLAT || op->op_type == OP_TRIL)
      need_zero_act_grads = true;                                                // these accumulate into their operands' gradients
    if (Linear* li = dynamic_cast<Linear*>(op)) {
      li->dx_overwrite = consumers[li->inputs[0].impl] == 1;
      // Linear(ReLU) -> Linear with nothing else reading the tensor in between: the upper layer applies the lower layer's
      // relu' to the gradient it hands down (FFH_LINEAR_DX_MASK_BY_X), the lower one takes it as is (DY_PREMASKED) --
      // reluBackward [ref: src/runtime/cuda_helper.cu:71-78] moved to where its operand is produced, so that no backward
      // kernel has to re-read y next to dy
      Linear* below = li->inputs[0].owner_op ? dynamic_cast<Linear*>(const_cast<Op*>(li->inputs[0].owner_op)) : nullptr;
      if (below && below->activation == AC_MODE_RELU && li->dx_overwrite && !li->discard_input_grad) {
        li->dx_mask_by_x = true;
        below->dy_premasked = true;
      }
      // ... and where the gradient this layer stores is the lower layer's final dy (no activation, or the ReLU whose derivative this
      // layer applies), the lower layer's bias gradient -- the column sums of that dy -- comes out of this layer's data-gradient
      // kernel (ffh_linear_bwd_set_dx_colsum) when the persistent kernel runs it; the lower layer's weight-gradient GEMM then runs
      // without the sums (6 % of it)
      li->colsum_lower = nullptr;
      if (config.dx_colsum && below && below->use_bias && li->dx_overwrite && !li->discard_input_grad && below->outputs[0].impl->pieces.empty() &&
          ((below->activation == AC_MODE_RELU && li->dx_mask_by_x) || below->activation == AC_MODE_NONE) && li->in_padded == li->in_channels)
        li->colsum_lower = below;
      // ... and two NARROW layers in a row (256 -> 64 -> 16 at the end of DLRM's bottom MLP): the upper layer's backward launch
      // also produces the lower layer's data gradient (ffh_linear_pair_bwd); shapes it does not serve fall back at run time
      li->pair_lower = nullptr;
      if (config.fuse_pair && below && li->dx_overwrite && !below->discard_input_grad && below->inputs[0].impl->pieces.empty() &&
          ((below->activation == AC_MODE_RELU && li->dx_mask_by_x) || below->activation == AC_MODE_NONE) && li->out_channels <= 16 &&
          (li->in_channels == 32 || li->in_channels == 64) && below->in_channels % 32 == 0)
        li->pair_lower = below;
      // the forward of such a pair needs less: the lower output read by the upper layer only, both outputs single buffers
      if (config.fuse_pair && below && consumers[li->inputs[0].impl] == 1 && li->out_channels <= 16 && (li->in_channels == 32 || li->in_channels == 64) &&
          below->inputs[0].impl->pieces.empty() && li->outputs[0].impl->pieces.empty() && li->layer_index == below->layer_index + 1)
        below->pair_upper = li;
    }
    if (DotInteraction* di = dynamic_cast<DotInteraction*>(op)) {
      di->bwd_overwrite = consumers[di->inputs[0].impl] == 1;
      if (!di->bwd_overwrite) need_zero_act_grads = true;
    }
    if (Concat* c = dynamic_cast<Concat*>(op)) {
      // inputs that nothing else reads take their gradient slice as a plain store (FFH_CONCAT_BWD_OVERWRITE)
      c->bwd_overwrite = true;
      for (int i = 0; i < c->numInputs; i++)
        if (consumers[c->inputs[i].impl] != 1) c->bwd_overwrite = false;
      for (int i = 0; i < c->numInputs; i++) {
        TensorImpl* im = c->inputs[i].impl;
        const bool via_exchange = exchange && c->inputs[i].owner_op && c->inputs[i].owner_op->op_type == OP_EMBEDDING &&
                                  !static_cast<const Embedding*>(c->inputs[i].owner_op)->row_sharded &&
                                  !static_cast<const Embedding*>(c->inputs[i].owner_op)->replicated;
        if (via_exchange && !c->bwd_overwrite) need_zero_gsend = true;
        if (im->grad && !im->grad_alias && !via_exchange && im->pieces.empty() && !c->bwd_overwrite) need_zero_act_grads = true;   // add_with_stride accumulates
      }
    }
  }

  // ---- 4c. the kernel that completes the embedding output gradients -----------------------------
  // In reverse layer order the tables come right after the Concat that gathers them; when that Concat's backward has
  // nothing to launch (every input writes its gradient slice in place) the op before it -- the first top-MLP layer --
  // produces those gradients, and the "gradients ready" event for the side-stream update can ride on its kernel.
  grad_attach_layer = -1;
  if (!embeddings.empty() && config.attach_events && config.overlap_embedding && !exchange && !use_workers() && fused_embedding_update()) {
    size_t l = (size_t)embeddings.back()->layer_index + 1;
    while (l < layers.size()) {
      Concat* c = dynamic_cast<Concat*>(layers[l]);
      if (!c) break;
      bool noop = true;
      for (int i = 0; i < c->numInputs; i++) {
        auto it = alias_of.find(c->inputs[i].impl);
        if (it == alias_of.end() || it->second.first != c) noop = false;
      }
      if (!noop) { l = layers.size(); break; }
      l++;
    }
    if (l < layers.size() && layers[l]->op_type == OP_LINEAR && l == (size_t)embeddings.back()->layer_index + 2) grad_attach_layer = (int)l;
  }

  // ---- 4c'. the last forked weight-gradient GEMM that reads the buffer the tables are gathered into ----------
  // The next step's gather (side stream) overwrites embedding outputs that alias a Concat output; the Linear layers consuming
  // that output read it as the x operand of their weight-gradient GEMMs on dw_stream.  In backward order the lowest-index such
  // layer comes last: behind its backward the gather may go, without waiting for the bottom MLP's weight gradients.
  z_reader_layer = -1;
  if (!embeddings.empty() && !exchange) {
    std::set<const TensorImpl*> zs;
    bool known = true;
    for (const Embedding* e : embeddings) {
      auto it = alias_of.find(e->outputs[0].impl);
      if (it == alias_of.end()) { known = false; break; }     // a table with storage of its own: some other op reads it -- keep the full join
      zs.insert(it->second.first->outputs[0].impl);
    }
    if (known) {
      for (size_t l = 0; l < layers.size(); l++) {
        if (layers[l]->op_type != OP_LINEAR) continue;
        if (zs.count(layers[l]->inputs[0].impl)) { z_reader_layer = (int)l; break; }
      }
    }
  }

  // the Linear with the most multiply-adds: its weight-gradient GEMM gets dw_stream to itself (Linear::backward_part)
  big_dw_layer = -1;
  {
    double best = 0.0;
    for (size_t l = 0; l < layers.size(); l++) {
      const Linear* li = layers[l]->op_type == OP_LINEAR ? static_cast<const Linear*>(layers[l]) : nullptr;
      if (!li) continue;
      const double m = (double)li->in_channels * li->out_channels;
      if (m > best) { best = m; big_dw_layer = (int)l; }
    }
  }

  // ---- 4d. exchange mode: the feature Concat's backward folded into the layer above it ----------
  // There the embedding gradients have to reach the all-to-all send buffer, which Concat::backward does with a pack
  // kernel on the critical stream.  The Linear that consumes the Concat can store each column of its data gradient where
  // that kernel would copy it (ffh_linear_bwd_set_dx_scatter); the column -> (buffer, leading dimension) map is fixed here.
  scatter_attach_layer = -1;
  for (Op* op : layers) {
    Linear* li = dynamic_cast<Linear*>(op);
    if (!li) continue;
    Concat* c = li->inputs[0].owner_op ? dynamic_cast<Concat*>(const_cast<Op*>(li->inputs[0].owner_op)) : nullptr;
    if (!exchange || !config.dx_scatter || !c || c->axis != 0 || !li->dx_overwrite || !c->bwd_overwrite || li->discard_input_grad) continue;
    std::vector<ffh_col_dest> map;
    bool ok = true;
    for (int i = 0; i < c->numInputs && ok; i++) {
      const TensorImpl* im = c->inputs[i].impl;
      if (!im->pieces.empty()) {
        for (const TensorPiece& pc : im->pieces)
          for (int64_t k = 0; k < pc.cols; k++) map.push_back({pc.grad + k, pc.ld});
      } else if (im->grad) {
        for (int k = 0; k < c->inputs[i].adim[0]; k++) map.push_back({im->grad + k, im->grad_ld});
      } else {
        ok = false;
      }
    }
    if (!ok || (int)map.size() != li->in_channels) continue;
    li->dx_map = (ffh_col_dest*)dmalloc(map.size() * sizeof(ffh_col_dest));
    check(api->ffh_memcpy_h2d(ctx, li->dx_map, map.data(), map.size() * sizeof(ffh_col_dest), stream), "dx map");
    check(api->ffh_stream_sync(ctx, stream), "dx map");
    li->dx_map_concat = c;
    // the tables come right after this Concat in reverse order: the scattered dX also completes their gradients
    if (!embeddings.empty() && config.attach_events && config.overlap_embedding && fused_embedding_update() &&
        c->layer_index == embeddings.back()->layer_index + 1 && li->layer_index == c->layer_index + 1)
      scatter_attach_layer = li->layer_index;
  }

  // ---- 4e. chains of narrow Linear layers ------------------------------------------------------------------------------
  // A run of consecutive Linear layers, each the only reader of the one below, every width <= FFH_CHAIN_MAX_WIDTH (the bottom MLP
  // 13-512-256-128; the Kaggle shape's 13-512-256-64-16 and 432-512-256-1): one launch forward (the lowest member's forward()), and for
  // the backward one call on the top member (FFModel::backward) -- at 2048-8192 samples per GPU these layers are 5-30 us kernels that wait
  // for each other, ~12 launches and ~100 us of the 1.18 ms per-rank step.  The backward chain leaves out the
  // model's last layer (the loss step is folded into its own one-launch backward) and a lowest member whose data gradient goes through
  // the exchange path's column map.
  for (Op* op : layers)
    if (Linear* li = dynamic_cast<Linear*>(op)) { li->chain_fwd.clear(); li->chain_bwd.clear(); li->fwd_done_by_chain = false; }
  if (config.mlp_chain && !config.profiling && !config.async_launch && !config.allow_tensor_op_math_conversion && !config.fp32_split_bf16x3) {
    auto member_ok = [&](const Linear* li) {
      return li->in_channels <= FFH_CHAIN_MAX_WIDTH && li->out_channels <= FFH_CHAIN_MAX_WIDTH && li->in_padded == li->in_channels &&
             li->inputs[0].impl->pieces.empty() && li->outputs[0].impl->pieces.empty() && li->inputs[0].impl->ptr && li->outputs[0].impl->ptr;
    };
    size_t l = 0;
    while (l < layers.size()) {
      Linear* a = layers[l]->op_type == OP_LINEAR ? static_cast<Linear*>(layers[l]) : nullptr;
      if (!a || !member_ok(a)) { l++; continue; }
      std::vector<Linear*> ch{a};
      while (l + ch.size() < layers.size() && ch.size() < (size_t)FFH_CHAIN_MAX_LAYERS) {
        Op* nx = layers[l + ch.size()];
        Linear* b = nx->op_type == OP_LINEAR ? static_cast<Linear*>(nx) : nullptr;
        Linear* lo = ch.back();
        if (!b || !member_ok(b) || b->inputs[0].impl != lo->outputs[0].impl || consumers[lo->outputs[0].impl] != 1) break;
        ch.push_back(b);
      }
      l += ch.size();
      if (ch.size() < 2) continue;
      int64_t nweights = 0;
      for (Linear* m : ch) nweights += (int64_t)m->in_channels * m->out_channels;
      if (nweights > config.mlp_chain_max_weights) continue;
      if (ch.size() >= 3) a->chain_fwd = ch;             // (two layers: one launch saved does not pay for the weights every CU streams)
      std::vector<Linear*> bw = ch;
      if (bw.back() == layers.back()) bw.pop_back();
      if (!bw.empty() && bw.front()->dx_map) bw.erase(bw.begin());
      bool ok = bw.size() >= 2;
      for (size_t i = 0; ok && i < bw.size(); i++) {
        const ActiMode am = bw[i]->activation;
        if (i + 1 < bw.size()) ok = (am == AC_MODE_NONE || am == AC_MODE_RELU) && bw[i + 1]->dx_overwrite && !bw[i + 1]->discard_input_grad;
        else ok = am == AC_MODE_NONE || am == AC_MODE_RELU || am == AC_MODE_SIGMOID;
      }
      if (ok) bw.back()->chain_bwd = bw;
      // (the two-narrow-layers launches of the same layers keep their pointers: where a chain call is not usable -- the batch -- they
      //  serve as before; where it is, the chain's lowest / top member is reached first and marks the others done)
    }
  }

  // ---- 5. parameters: one slab for every Linear tensor, tables on their own ---------------------
  // (a Linear kernel whose input was padded in step 4a is [out][in_padded] here: pad columns zero, and kept zero by every optimizer --
  //  their gradient is dy^T times the input's zero pad columns)
  auto slab_elems = [&](const Parameter& p) -> size_t {
    if (p.owner_op->op_type == OP_LINEAR && p.numDim == 2) return (size_t)p.adim[1] * (size_t)static_cast<const Linear*>(p.owner_op)->in_padded;
    return p.get_volume();
  };
  // (a tensor's range is a whole number of 32 floats: every tensor starts a 128-byte line and a group of the split mode's plane image,
  //  include/ff_hip.h ffh_ctx_bf16x3_mirror_set; the pad floats are zeros that every slab-wise kernel keeps zero)
  auto slab_span = [&](const Parameter& p) -> size_t { return (slab_elems(p) + 31) / 32 * 32; };
  mlp_count = 0;
  for (Parameter& p : parameters)
    if (in_dense_slab(p)) mlp_count += slab_span(p);
  mlp_weights = (float*)dmalloc(std::max<size_t>(mlp_count, 64) * 4);
  mlp_grads = (float*)dmalloc(std::max<size_t>(mlp_count, 64) * 4);
  check(api->ffh_zero(ctx, mlp_weights, std::max<size_t>(mlp_count, 64) * 4, stream), "zero");
  check(api->ffh_zero(ctx, mlp_grads, std::max<size_t>(mlp_count, 64) * 4, stream), "zero");
  size_t off_p = 0;
  const bool fused = fused_embedding_update();
  for (Parameter& p : parameters) {
    TensorImpl* im = p.impl;
    im->ld = p.adim[0];
    im->rows_local = (int64_t)(p.get_volume() / (size_t)p.adim[0]);
    if (in_dense_slab(p)) {
      if (p.owner_op->op_type == OP_LINEAR && p.numDim == 2) im->ld = static_cast<const Linear*>(p.owner_op)->in_padded;
      im->ptr = mlp_weights + off_p;
      im->grad = mlp_grads + off_p;
      im->grad_ld = im->ld;
      im->alias = true;
      im->bytes = slab_elems(p) * 4;
      off_p += slab_span(p);
    } else {
      Embedding* e = static_cast<Embedding*>(p.owner_op);
      if (!e->held_here(rank)) continue;   // sole owner (or one column / row block per rank): never replicated, never all-reduced
      im->bytes = p.get_volume() * 4;
      im->ptr = dmalloc(im->bytes + (e->row_sharded ? (size_t)e->out_channels * 4 : 0));   // row block: + the zero row
      if (e->row_sharded) check(api->ffh_zero(ctx, (char*)im->ptr + im->bytes, (size_t)e->out_channels * 4, stream), "zero row");
      if (!fused) {
        im->grad = (float*)dmalloc(im->bytes + (e->row_sharded ? (size_t)e->out_channels * 4 : 0));   // row block: foreign ids pile onto the zero row
        im->grad_ld = im->ld;
      }
    }
  }
  // Op::weights[] are copies of the Parameters: same impl pointers, nothing to patch.

  // ---- 5a'. scratch of the direct all-reduce (--direct-allreduce): the received slices + the gathered sums of the largest range ----
  if (ar_scratch) { api->ffh_free(ctx, ar_scratch); ar_scratch = nullptr; ar_scratch_floats = 0; }
  if (config.direct_allreduce && exchange && mlp_count) {
    const int64_t slice = ((((int64_t)mlp_count + world_size - 1) / world_size) + 3) / 4 * 4;
    ar_scratch_floats = (size_t)(2 * slice * world_size);
    ar_scratch = (float*)dmalloc(ar_scratch_floats * 4);
  }

  // ---- 5b. buckets of the MLP gradients' all-reduce ---------------------------------------------------------------------
  // In the reference every parameter has its own update task with its own ncclAllReduce, ordered by region dependences only: a top
  // layer's gradients are summed over the ranks while the layers below still run their backward [ref: src/runtime/optimizer.cc:93-189,
  // src/runtime/model.cc:1471-1477].  Here: the Linear layers' slab ranges, walked in backward order and merged until a bucket holds
  // allreduce_bucket_floats gradients; a bucket is issued on ar_stream as soon as the layers it covers have issued their backward
  // (FFModel::issue_grad_buckets), the slab optimizer waits for all of them.  What no bucket covers (data-parallel tables in the slab)
  // is reduced in update() as before.
  for (GradBucket& b : grad_buckets) { api->ffh_event_destroy(ctx, b.ready); api->ffh_event_destroy(ctx, b.ready_dw); api->ffh_event_destroy(ctx, b.done); }
  grad_buckets.clear();
  grad_rest.clear();
  if (exchange && mlp_count) {
    // (row blocks of the biggest layer: measured on one GPU -- forced 1-rank RCCL exchange, 4096 samples -- every extra block costs the
    //  step ~25 us (1.297 / 1.331 / 1.365 ms at 1 / 2 / 4 blocks: four GEMMs of a quarter of the rows take 268 us where one takes 207); what it
    //  would hide of a 14 MB ring all-reduce could not be measured without a multi-GPU box: off unless asked for)
    int chunks = config.big_dw_chunks > 0 ? config.big_dw_chunks : 1;
    Linear* big = big_dw_layer >= 0 ? static_cast<Linear*>(layers[big_dw_layer]) : nullptr;
    // (cut only a layer worth cutting whose dy is final when its backward starts, into row blocks whose dy column slices stay 16-byte aligned)
    if (!big || chunks < 2 || (int64_t)big->in_channels * big->out_channels < config.big_dw_min_weights || big->out_channels % (4 * chunks) != 0 ||
        !(big->dy_premasked || big->activation == AC_MODE_NONE) || big->discard_input_grad || !config.parallel_dw)
      chunks = 1;
    auto build = [&](size_t threshold) {
      std::vector<GradBucket> out;
      GradBucket cur{0, 0, -1, false, nullptr, nullptr, nullptr, -1, 0, false};
      auto close = [&]() { if (cur.count) out.push_back(cur); cur = GradBucket{0, 0, -1, false, nullptr, nullptr, nullptr, -1, 0, false}; };
      for (int l = (int)layers.size() - 1; l >= 0; l--) {
        Linear* li = layers[l]->op_type == OP_LINEAR ? static_cast<Linear*>(layers[l]) : nullptr;
        if (!li) continue;
        // the layer's range: kernel, then bias, adjacent in the slab (step 5)
        const size_t lo = (size_t)(li->weights[0].impl->grad - mlp_grads);
        size_t hi = lo + (li->weights[0].impl->bytes / 4 + 31) / 32 * 32;
        if (li->use_bias) hi = (size_t)(li->weights[1].impl->grad - mlp_grads) + (li->weights[1].impl->bytes / 4 + 31) / 32 * 32;
        if (li == big && chunks > 1) {      // its own buckets: row block c of the kernel; the last one takes the bias too
          close();
          const size_t per = (size_t)(big->out_channels / chunks) * (size_t)big->weights[0].impl->ld;
          for (int c = 0; c < chunks; c++) {
            GradBucket b{lo + c * per, c == chunks - 1 ? hi - (lo + c * per) : per, l, false, nullptr, nullptr, nullptr, l, c, false};
            out.push_back(b);
          }
          continue;
        }
        if (cur.count && hi != cur.off) close();          // not adjacent to the bucket being filled (tables in between)
        if (!cur.count) { cur.off = lo; cur.count = hi - lo; }
        else { cur.count += cur.off - lo; cur.off = lo; }
        cur.lowest_layer = l;
        if (cur.count >= threshold) close();
      }
      close();
      // a small tail (the bottom MLP behind the biggest layer) joins the bucket before it where the two are adjacent: both wait for the
      // last weight-gradient GEMM anyway, and one call fewer stands between it and the optimizer
      if (out.size() >= 2) {
        GradBucket& t = out.back();
        GradBucket& p = out[out.size() - 2];
        if (t.count < threshold / 4 && p.chunk_layer < 0 && t.off + t.count == p.off) { p.off = t.off; p.count += t.count; p.lowest_layer = t.lowest_layer; out.pop_back(); }
      }
      return out;
    };
    size_t threshold = (size_t)std::max<int64_t>(config.allreduce_bucket_floats, 1);
    grad_buckets = build(threshold);
    while (grad_buckets.size() > 8 && threshold < mlp_count) { threshold *= 2; grad_buckets = build(threshold); }     // (the probes number eight)
    std::vector<std::pair<size_t, size_t>> covered;
    for (GradBucket& b : grad_buckets) {
      for (ffh_event* e : {&b.ready, &b.ready_dw, &b.done}) check(api->ffh_event_create(ctx, e), "event create");
      covered.push_back({b.off, b.count});
    }
    std::sort(covered.begin(), covered.end());
    size_t at = 0;
    for (auto& c : covered) { if (c.first > at) grad_rest.push_back({at, c.first - at}); at = c.first + c.second; }
    if (at < mlp_count) grad_rest.push_back({at, mlp_count - at});
  }

  // ---- 6. workspace + metrics -------------------------------------------------------------------
  workspace_bytes = 256;
  if (owned_shards && fused) {
    const int chunk = std::min(owned_shards, FFH_MAX_TABLES);
    workspace_bytes = api->ffh_embedding_bwd_workspace_bytes(chunk, L, D, config.batchSize) + 256;
  }
  for (Embedding* e : embeddings)
    if (e->row_sharded) workspace_bytes = std::max(workspace_bytes, api->ffh_embedding_bwd_workspace_bytes(1, L, D, config.batchSize) + 256);
  workspace = dmalloc(workspace_bytes);
  check(api->ffh_ctx_set_workspace(ctx, workspace, workspace_bytes), "set workspace");
  int n_replicated = 0;
  for (Embedding* e : embeddings) n_replicated += e->replicated;
  repl_workspace = nullptr; repl_workspace_bytes = 0;
  if (n_replicated) {
    repl_workspace_bytes = api->ffh_embedding_bwd_workspace_bytes(std::min(n_replicated, FFH_MAX_TABLES), L, D, local_batch) + 256;
    repl_workspace = dmalloc(repl_workspace_bytes);
  }
  if (side_worker) check(api->ffh_ctx_set_workspace(side_worker->ctx(), workspace, workspace_bytes), "set workspace");   // the only other user
  layer_events.resize(layers.size(), nullptr);
  for (ffh_event& e : layer_events) check(api->ffh_event_create(ctx, &e), "event create");
  d_perf = (ffh_perf_metrics*)dmalloc(sizeof(ffh_perf_metrics));
  check(api->ffh_zero(ctx, d_perf, sizeof(ffh_perf_metrics), stream), "zero");
  check(api->ffh_zero(ctx, act_slab, std::max<size_t>(act_bytes, 256), stream), "zero");
  check(api->ffh_zero(ctx, act_grad_slab, std::max<size_t>(act_grad_bytes, 256), stream), "zero");

  // ---- 7. tensor-op math mode: bf16 twins ---------------------------------------------------------
  // [ref: --allow-tensor-op-math-conversion, src/runtime/model.cu:81-83]  The library rounds GEMM operands to bfloat16 in that
  // mode; a buffer whose EVERY writer keeps a bf16 twin current can be read at half the bytes (include/ff_hip.h,
  // ffh_ctx_bf16_mirror_set: validity is this layer's contract).  Twin-writers: a Linear with in, out >= FFH_BF16_MIN_DIM
  // (forward: its output; backward: its input gradient, when it is that gradient's only producer), the gather (embedding
  // outputs of a width divisible by 4), the slab optimizer (weights).  Registered:
  //   * the weight slab (reconverted here whenever the host or an initializer wrote weights);
  //   * the output of such a Linear with storage of its own; a Concat output all of whose inputs are written in place by
  //     tables and such Linears;
  //   * the gradient buffer of a tensor whose single consumer is such a Linear storing (not accumulating) its data gradient.
  // The split mode (--fp32-split-bf16x3) keeps, for the same buffers and by the same rules, the THREE-PLANE IMAGE of ffh_ctx_bf16x3_mirror_set
  // (6 bytes per element; FFH_BF16X3_IMAGE_BYTES): its GEMMs then stream the operands' bf16 terms by LDS-DMA instead of splitting fp32 tiles in
  // registers (csrc/linear_x3_dma.hip).  twin_at() below is the one place the two layouts differ for this layer.
  n_twin_regions = 0;
  for (Op* op : layers) if (op->op_type == OP_LINEAR) static_cast<Linear*>(op)->dx_image = false;
  const bool x3_images = config.fp32_split_bf16x3 && !config.allow_tensor_op_math_conversion;
  if ((config.allow_tensor_op_math_conversion || x3_images) && config.bf16_twins && mlp_count > 0) {
    const size_t ab = std::max<size_t>(act_bytes, 256);
    const size_t act_tb = x3_images ? FFH_BF16X3_IMAGE_BYTES(ab) : ab / 2, w_tb = x3_images ? FFH_BF16X3_IMAGE_BYTES((size_t)mlp_count * 4) : (size_t)mlp_count * 2;
    act_twin = dmalloc(act_tb + 256); grad_twin = dmalloc(act_tb + 256); w_twin = dmalloc(w_tb + 256);
    check(api->ffh_zero(ctx, act_twin, act_tb + 256, stream), "zero"); check(api->ffh_zero(ctx, grad_twin, act_tb + 256, stream), "zero");
    check(api->ffh_zero(ctx, w_twin, w_tb + 256, stream), "zero");
    // the twin / image address of the fp32 byte offset `off` of a slab (the image: whole 128-byte groups only)
    auto twin_at = [&](void* twin_base, size_t off) -> void* {
      if (!x3_images) return (char*)twin_base + off / 2;
      return off % 128 ? nullptr : (char*)twin_base + off / 128 * 192;
    };
    auto reg = [&](const void* base, size_t bytes, void* twin) {
      if (bytes == 0 || !twin || n_twin_regions >= 30) return;
      auto set = x3_images ? api->ffh_ctx_bf16x3_mirror_set : api->ffh_ctx_bf16_mirror_set;
      check(set(ctx, base, bytes, twin), "bf16 twin");
      if (dw_worker) check(set(dw_worker->ctx(), base, bytes, twin), "bf16 twin");
      if (side_worker) check(set(side_worker->ctx(), base, bytes, twin), "bf16 twin");
      n_twin_regions++;
    };
    reg(mlp_weights, (size_t)mlp_count * 4, w_twin);
    w_twin_dirty = true;
    auto twin_linear = [&](const Op* op) {
      const Linear* l = op && op->op_type == OP_LINEAR ? static_cast<const Linear*>(op) : nullptr;
      return l && l->in_channels >= FFH_BF16_MIN_DIM && l->out_channels >= FFH_BF16_MIN_DIM &&
             (!x3_images || 2.0 * (double)local_rows(l->outputs[0], this) * (double)l->in_padded * (double)l->out_channels >= FFH_BF16X3_MIN_FLOP);      // (the layers the mode takes: include/ff_hip.h)
    };
    auto in_slab = [&](const void* q) { return (const char*)q >= act_slab && (const char*)q < act_slab + act_bytes; };
    for (Op* op : layers) {
      TensorImpl* im = op->outputs[0].impl;
      if (!im || !im->ptr || alias_of.count(im) || !in_slab(im->ptr) || !im->pieces.empty()) continue;
      bool act_ok = false;
      if (twin_linear(op)) act_ok = true;
      else if (Concat* c = dynamic_cast<Concat*>(op)) {
        act_ok = !exchange && c->numInputs > 0 && (!x3_images || im->ld % 32 == 0);      // (the gather writes the image of rows that are whole 32-element groups apart)
        std::vector<int> x3_convert;
        c->image_inputs.clear();
        for (int i = 0; i < c->numInputs && act_ok; i++) {
          const Tensor& in = c->inputs[i];
          auto it = alias_of.find(in.impl);
          if (it == alias_of.end() || it->second.first != c || !in.owner_op) { act_ok = false; break; }
          if (in.owner_op->op_type == OP_EMBEDDING) act_ok = static_cast<const Embedding*>(in.owner_op)->out_channels % 4 == 0 && !static_cast<const Embedding*>(in.owner_op)->replicated;
          else if (x3_images && in.owner_op->op_type == OP_LINEAR && !twin_linear(in.owner_op) && !use_workers()) x3_convert.push_back(i);   // its slice's image in Concat::forward
          else act_ok = twin_linear(in.owner_op);
        }
        if (act_ok) c->image_inputs = x3_convert;
      }
      const Linear* only = nullptr; int ncons = 0;
      for (Op* q : layers)
        for (int i = 0; i < q->numInputs; i++)
          if (q->inputs[i].impl == im) { ncons++; only = q->op_type == OP_LINEAR ? static_cast<const Linear*>(q) : nullptr; }
      // (round 5) a Linear the bf16 pipe does not take (13 -> 512 under the bottom MLP) whose only reader is one it does take: the twin by an
      // explicit conversion behind its forward call -- 100 MB of traffic at 32768 samples, for which the reader's forward and weight
      // gradient take both operands from twins (the LDS-DMA kernels instead of the converting 128 x 128 one)
      if (!act_ok && config.bf16_convert_twins && op->op_type == OP_LINEAR && ncons == 1 && only && twin_linear(only) && !use_workers() && im->ld % 8 == 0) {
        Linear* li = static_cast<Linear*>(op);
        if (!li->pair_upper && !li->pair_lower) {      // (the chain launches stand back in tensor-op mode: mlp_chain_usable)
          li->out_twin = twin_at(act_twin, (size_t)((const char*)im->ptr - act_slab));
          li->out_twin_x3 = x3_images;
          act_ok = li->out_twin != nullptr;
        }
      }
      if (act_ok) reg(im->ptr, im->bytes, twin_at(act_twin, (size_t)((const char*)im->ptr - act_slab)));
      // the gradient of this tensor: one consumer, a twin-writing Linear that stores its data gradient
      // (... or the one-launch backward of a layer with <= 4 outputs, which writes the twin of its data gradient too: the
      //  256 -> 1 layer on top of the Terabyte MLP, whose input gradient is the 512 -> 256 layer's dy)
      auto skinny_twin = [&](const Linear* l) {
        return l && l->out_channels <= 4 && l->in_channels >= FFH_BF16_MIN_DIM && l->in_channels <= 1024 && l->in_channels % 4 == 0 && !config.deterministic;
      };
      // ... and only where somebody reads that twin: the producer of the tensor is a twin-reading Linear whose dy arrives final
      // (premasked by the consumer's dX epilogue, or no activation).  The Concat output's gradient (3456 columns at the Terabyte
      // shape, of which the bottom MLP reads 128 through a live relu') has no such reader: 226 MB per step not written
      const Linear* prod = op->op_type == OP_LINEAR ? static_cast<const Linear*>(op) : nullptr;
      const bool twin_read = prod && twin_linear(prod) && (prod->dy_premasked || prod->activation == AC_MODE_NONE);
      if (ncons == 1 && twin_read && only && (twin_linear(only) || skinny_twin(only)) && only->dx_overwrite && !only->discard_input_grad && im->grad && !exchange)
        reg(im->grad, im->bytes, twin_at(grad_twin, (size_t)((const char*)im->grad - act_grad_slab)));
      // split mode: the consumer is a Linear on the fp32 kernels (below FFH_BF16X3_MIN_WEIGHTS: 512 -> 256 on top of the Terabyte MLP) -- the image of
      // the data gradient it stores by a pass behind its backward call (Linear::backward_part), so that the layer this gradient is the dy of
      // (1024 -> 512) streams it: 60 us of conversion for 170 us of split-in-kernel GEMM at 32768 samples
      else if (x3_images && ncons == 1 && twin_read && only && only->dx_overwrite && !only->discard_input_grad && im->grad && !exchange && !use_workers() &&
               !only->pair_upper && !only->pair_lower && twin_at(grad_twin, (size_t)((const char*)im->grad - act_grad_slab))) {
        reg(im->grad, im->bytes, twin_at(grad_twin, (size_t)((const char*)im->grad - act_grad_slab)));
        const_cast<Linear*>(only)->dx_image = true;
      }
    }
  }
  check(api->ffh_stream_sync(ctx, stream), "allocate sync");
}

void FFModel::init_layers() {
  if (!compiled) die("init_layers() before compile()");
  for (Op* op : layers) op->init(*this);
}

void FFModel::print_layers(int id) {
  if (id == -1) for (Op* op : layers) op->print_layer(*this);
  else layers.at(id)->print_layer(*this);
}

// =============================================================================================
// embedding group: batched gather (+ exchange) and batched fused update
// =============================================================================================
// launches one batched kernel per distinct shard width (all table-wise tables share one; column blocks of
// giant tables another); FWD: gather, else fused backward + SGD
enum ShardLaunch { kGather, kFusedUpdate, kSortOnly, kApplyOnly };
static void launch_shard_groups(const FFModel* ff, ShardLaunch what, ffh_stream s, ffh_ctx* cx, const std::vector<const int64_t*>* idx_override = nullptr) {
  const bool fwd = what == kGather;
  ffh_sparse_opt rule;
  const bool ruled = (what == kFusedUpdate || what == kApplyOnly) && ff->sparse_rule(rule);     // momentum / wd SGD, Adam on the touched rows
  const int L = ff->embeddings[0]->inputs[0].adim[0];
  const int aggr = (int)ff->embeddings[0]->aggr;
  std::map<int, std::vector<ffh_emb_table>> by_cols;
  std::map<int, std::vector<ffh_emb_state>> st_by_cols;
  size_t owned_i = 0;
  for (const FFModel::EmbShard& sh : ff->shards) {
    if (sh.owner != ff->rank) continue;
    const Embedding* e = sh.e;
    ffh_emb_table t;
    t.idx = (const int64_t*)e->inputs[0].impl->ptr;
    if (idx_override && owned_i < idx_override->size()) t.idx = (*idx_override)[owned_i];
    owned_i++;
    t.weight = (float*)e->weights[0].impl->ptr;      // column-sharded: the local [R][cols] slice
    t.num_entries = e->num_entries;
    if (!ff->exchange) {
      t.io = fwd ? (float*)e->outputs[0].impl->ptr : e->outputs[0].impl->grad;
      t.ld = fwd ? e->outputs[0].impl->ld : e->outputs[0].impl->grad_ld;
    } else {
      t.io = (fwd ? ff->xsend : ff->grecv) + sh.off;
      t.ld = ff->rank_width[ff->rank];
    }
    by_cols[sh.cols].push_back(t);
    st_by_cols[sh.cols].push_back(ffh_emb_state{e->opt_state[0], e->opt_state[1]});
  }
  for (auto& kv : by_cols) {
    std::vector<ffh_emb_table>& tabs = kv.second;
    const std::vector<ffh_emb_state>& sts = st_by_cols[kv.first];
    for (size_t b = 0; b < tabs.size(); b += FFH_MAX_TABLES) {
      const int n = (int)std::min<size_t>(FFH_MAX_TABLES, tabs.size() - b);
      const int64_t B = ff->config.batchSize;
      switch (what) {
        case kGather: ff->check(ff->api->ffh_embedding_fwd_multi(cx, tabs.data() + b, n, L, kv.first, B, aggr, s), "embedding_fwd_multi"); break;
        case kFusedUpdate:
          if (ruled) ff->check(ff->api->ffh_embedding_bwd_opt_fused_multi(cx, tabs.data() + b, sts.data() + b, n, L, kv.first, B, aggr, &rule, s), "embedding_bwd_opt_fused_multi");
          else ff->check(ff->api->ffh_embedding_bwd_sgd_fused_multi(cx, tabs.data() + b, n, L, kv.first, B, aggr, rule.lr, s), "embedding_bwd_sgd_fused_multi");
          break;
        case kSortOnly: ff->check(ff->api->ffh_embedding_bwd_sort_multi(cx, tabs.data() + b, n, L, kv.first, B, s), "embedding_bwd_sort_multi"); break;
        case kApplyOnly:
          if (ruled) ff->check(ff->api->ffh_embedding_bwd_opt_apply_multi(cx, tabs.data() + b, sts.data() + b, n, L, kv.first, B, aggr, &rule, s), "embedding_bwd_opt_apply_multi");
          else ff->check(ff->api->ffh_embedding_bwd_sgd_apply_multi(cx, tabs.data() + b, n, L, kv.first, B, aggr, rule.lr, s), "embedding_bwd_sgd_apply_multi");
          break;
      }
    }
  }
}
// the batched gather (fwd) or fused update kernels of this rank's shards alone, no exchange: what bench.py times as the
// roofline kernels of a multi-rank job
void FFModel::embedding_kernels_only(bool fwd, ffh_stream s, const std::vector<const int64_t*>* idx_override) const {
  if (embeddings.empty() || (!fwd && !fused_embedding_update())) return;
  launch_shard_groups(this, fwd ? kGather : kFusedUpdate, s, ctx, idx_override);
}

// The sort of the fused update reads only the sparse ids, which are final when the gather starts: issued behind the gather on
// the side stream it runs beside the top MLP's forward instead of between the gradients and the next gather (what Legion's
// region dependences would give an index-only task).  The sorted list waits in the ctx workspace, so this is only done when
// nothing else writes the workspace between a step's gather and its update: one launch group (one shard width, <=
// FFH_MAX_TABLES shards), no row-wise sharded table (its own fused call), launches issued inline.
bool FFModel::early_sort_possible(int where) const {
  if (!config.early_sort || !config.overlap_embedding || !fused_embedding_update() || config.profiling) return false;
  if (config.computationMode != COMP_MODE_TRAINING || use_workers()) return false;
  // by shape (round 4, profiles/r04_ab_schedule.txt): behind the exchange the whole update sits between the backward all-to-all and
  // the next gather, so the sort leaves that chain; on one GPU it pays at small per-GPU batches (4096 samples: 1.178 vs 1.191 ms)
  // and costs at large ones, where it runs beside the top MLP's first forward GEMM (32768: 7.76-7.79 vs 7.71-7.73; 8192, MLPerf
  // shape: 1.236-1.239 vs 1.227-1.233)
  const int mode = config.early_sort > 0 ? config.early_sort : ((!exchange && local_batch >= 8192) ? early_sort_big_batch_mode : 1);
  if (mode != where) return false;
  int n = 0, cols = -1;
  for (const EmbShard& sh : shards) {
    if (sh.owner != rank) continue;
    if (cols >= 0 && sh.cols != cols) return false;
    cols = sh.cols;
    n++;
  }
  for (const Embedding* e : embeddings)
    if (e->row_sharded) return false;
  return n > 0 && n <= FFH_MAX_TABLES;
}

void FFModel::probe_record(int which, ffh_stream s, ffh_ctx* cx) const {
  if (!probe_events_on) return;
  if (!probe_ev[which]) check(api->ffh_event_create(ctx, &probe_ev[which]), "probe event");
  check(api->ffh_event_record(cx, probe_ev[which], s), "probe event");
}

void FFModel::embedding_group_forward(ffh_stream s, ffh_ctx* on_ctx) const {
  if (embeddings.empty()) return;
  launch_shard_groups(this, kGather, s, on_ctx ? on_ctx : ctx);
  ffh_ctx* cx = on_ctx ? on_ctx : ctx;
  if (exchange) {
    // each owner gathered its tables / column blocks for the global batch; rows go to the rank that owns the sample
    probe_record(4, s, cx);
    if (!shards.empty() && config.comm.alltoall_f32(config.comm.user, xsend, fwd_send_counts.data(), xrecv, fwd_recv_counts.data(), s) != 0)
      die("alltoall (embedding forward) failed");
    probe_record(5, s, cx);
  }
  // data-parallel (replicated) tables: this rank's samples from this rank's copy, straight into the outputs; one launch
  {
    std::vector<ffh_emb_table> tabs;
    const int Lr = embeddings[0]->inputs[0].adim[0];
    for (const Embedding* e : embeddings) {
      if (!e->replicated) continue;
      ffh_emb_table t;
      t.idx = (const int64_t*)e->inputs[0].impl->ptr + (int64_t)rank * local_batch * Lr;   // every rank holds the ids of the global batch
      t.weight = (float*)e->weights[0].impl->ptr;
      t.num_entries = e->num_entries;
      t.io = (float*)e->outputs[0].impl->ptr;
      t.ld = e->outputs[0].impl->ld;
      tabs.push_back(t);
    }
    for (size_t b = 0; b < tabs.size(); b += FFH_MAX_TABLES) {
      const int n = (int)std::min<size_t>(FFH_MAX_TABLES, tabs.size() - b);
      check(api->ffh_embedding_fwd_multi(cx, tabs.data() + b, n, Lr, embeddings[0]->out_channels, local_batch, (int)embeddings[0]->aggr, s),
            "embedding_fwd_multi (data-parallel tables)");
    }
  }
  // row-wise sharded tables: partial bag sums of the GLOBAL batch over the rows held here (rows held elsewhere read the
  // zero row), then the ranks' partials are added and every rank keeps its own samples
  for (const Embedding* e : embeddings) {
    if (!e->row_sharded) continue;
    const int L = e->inputs[0].adim[0], D = e->out_channels;
    check(api->ffh_embedding_localize_rows(cx, (const int64_t*)e->inputs[0].impl->ptr, e->local_idx, (int64_t)config.batchSize * L,
                                           e->row_begin, e->rows_local, s), "embedding_localize_rows");
    check(api->ffh_embedding_fwd(cx, e->local_idx, e->partial, (const float*)e->weights[0].impl->ptr, L, D, config.batchSize,
                                 e->rows_local + 1, D, (int)e->aggr, s), e->name);
    if (config.comm.reduce_scatter_sum_f32(config.comm.user, e->partial, (float*)e->outputs[0].impl->ptr, local_batch * D, s) != 0)
      die("reduce-scatter (row-sharded embedding forward) failed");
  }
}

// Gradient of the data-parallel tables: G[row] += sum of this rank's gradient rows that looked the row up.  The reference
// scatter-adds with atomics; these are the SMALL tables (3 ... a few thousand rows), where a rank's samples pile hundreds of
// adds onto one address and atomics serialise.  The fused sparse kernels already compute exactly these segmented sums
// (sort + reduce, W[row] -= lr * sum): pointed at the zeroed slab gradient with lr = -1 they leave G = 0 + sum -- no
// atomics, a fixed order.  They need the scratch workspace, which the side-stream update of the owned tables may be using:
// this call brings its own (workspace pointers are read at launch time, so switching between launches is safe).
void FFModel::replicated_embedding_grads() const {
  if (!repl_workspace) return;
  std::vector<ffh_emb_table> tabs;
  const int L = embeddings[0]->inputs[0].adim[0], D = embeddings[0]->out_channels;
  for (const Embedding* e : embeddings) {
    if (!e->replicated) continue;
    ffh_emb_table t;
    t.idx = (const int64_t*)e->inputs[0].impl->ptr + (int64_t)rank * local_batch * L;
    t.weight = e->weights[0].impl->grad;
    t.num_entries = e->num_entries;
    t.io = e->outputs[0].impl->grad;
    t.ld = e->outputs[0].impl->grad_ld;
    tabs.push_back(t);
  }
  check(api->ffh_ctx_set_workspace(ctx, repl_workspace, repl_workspace_bytes), "set workspace");
  for (size_t b = 0; b < tabs.size(); b += FFH_MAX_TABLES) {
    const int n = (int)std::min<size_t>(FFH_MAX_TABLES, tabs.size() - b);
    check(api->ffh_embedding_bwd_sgd_fused_multi(ctx, tabs.data() + b, n, L, D, local_batch, (int)embeddings[0]->aggr, -1.0f, stream),
          "embedding gradient (data-parallel tables)");
  }
  check(api->ffh_ctx_set_workspace(ctx, workspace, workspace_bytes), "set workspace");
}

void FFModel::embedding_group_update(ffh_stream s, ffh_ctx* on_ctx) const {
  if (embeddings.empty()) return;
  if (exchange) {
    // gradients of the rows go back to the owners (transposed exchange)
    probe_record(6, s, on_ctx ? on_ctx : ctx);
    if (!shards.empty() && config.comm.alltoall_f32(config.comm.user, gsend, fwd_recv_counts.data(), grecv, fwd_send_counts.data(), s) != 0)
      die("alltoall (embedding backward) failed");
    probe_record(7, s, on_ctx ? on_ctx : ctx);
    bwd_alltoall_issued = true;
  }
  launch_shard_groups(this, emb_sorted_early ? kApplyOnly : kFusedUpdate, s, on_ctx ? on_ctx : ctx);
  emb_sorted_early = false;
  // row-wise sharded tables: every rank needs the gradient rows of the global batch; the fused update then touches the
  // rows held here, and whatever the other ranks' rows piled onto the zero row is wiped
  ffh_ctx* cx = on_ctx ? on_ctx : ctx;
  ffh_sparse_opt rule;
  const bool ruled = sparse_rule(rule);
  for (const Embedding* e : embeddings) {
    if (!e->row_sharded) continue;
    const int L = e->inputs[0].adim[0], D = e->out_channels;
    if (config.comm.allgather_f32(config.comm.user, e->outputs[0].impl->grad, e->gfull, local_batch * D, s) != 0)
      die("all-gather (row-sharded embedding backward) failed");
    float* w = (float*)e->weights[0].impl->ptr;
    if (ruled) {
      const ffh_emb_table t{e->local_idx, w, e->gfull, e->rows_local + 1, D};
      const ffh_emb_state st{e->opt_state[0], e->opt_state[1]};
      check(api->ffh_embedding_bwd_opt_fused_multi(cx, &t, &st, 1, L, D, config.batchSize, (int)e->aggr, &rule, s), e->name);
    } else {
      check(api->ffh_embedding_bwd_sgd_fused(cx, e->local_idx, e->gfull, w, L, D, config.batchSize, e->rows_local + 1, D, (int)e->aggr, rule.lr, s), e->name);
    }
    check(api->ffh_zero(cx, w + e->rows_local * (int64_t)D, (size_t)D * 4, s), "zero row");   // (its optimizer state is never read for a row of the block)
  }
}

// The reference's own table update on the rank(s) that hold a table, for optimizers the fused update does not cover (default for
// momentum / weight-decay SGD and Adam): Op::zero_grad [ref: src/runtime/model.cc:466-490] (zero_gradients()), embed_backward
// [ref: src/ops/embedding.cu:192-217] into the owner-local dense gradient, then the optimizer's dense sweep with its dense per-table
// state [ref: src/runtime/optimizer.cc:93-189,256-330].  Multi-rank: the rows' gradients first travel back to the owners (the
// transposed all-to-all / the all-gather of a row-sharded table); a table has ONE holder per element, so nothing is all-reduced.
void FFModel::embedding_dense_update() const {
  if (embeddings.empty()) return;
  const int L = embeddings[0]->inputs[0].adim[0];
  const int aggr = (int)embeddings[0]->aggr;
  if (exchange) {
    if (!shards.empty() && config.comm.alltoall_f32(config.comm.user, gsend, fwd_recv_counts.data(), grecv, fwd_send_counts.data(), stream) != 0)
      die("alltoall (embedding backward) failed");
    for (const EmbShard& sh : shards) {
      if (sh.owner != rank) continue;
      const Embedding* e = sh.e;
      check(api->ffh_embedding_bwd_dense(ctx, (const int64_t*)e->inputs[0].impl->ptr, grecv + sh.off, e->weights[0].impl->grad, L, sh.cols,
                                         config.batchSize, e->num_entries, rank_width[rank], aggr, stream), e->name);
    }
    for (const Embedding* e : embeddings) {
      if (!e->row_sharded) continue;
      const int D = e->out_channels;
      if (config.comm.allgather_f32(config.comm.user, e->outputs[0].impl->grad, e->gfull, local_batch * D, stream) != 0)
        die("all-gather (row-sharded embedding backward) failed");
      check(api->ffh_embedding_bwd_dense(ctx, e->local_idx, e->gfull, e->weights[0].impl->grad, L, D, config.batchSize, e->rows_local + 1, D, aggr, stream), e->name);
    }
  }
  for (Embedding* e : embeddings)
    if (e->held_here(rank) && !e->replicated) optimizer->update(&e->weights[0]);
}

// =============================================================================================
// the training step [ref: src/runtime/model.cc:1410-1477, examples/cpp/DLRM/dlrm.cc:166-182]
// =============================================================================================
// ---- bucketed all-reduce of the MLP gradients (allocate step 5b) ----------------------------------------------------------------
// The sum of a gradient range over the ranks.  Ring: the transport's all-reduce (ncclAllReduce: 2 (N - 1) / N of the bytes over ONE link per
// hop).  Direct (--direct-allreduce; SURVEY section 5: "prefer direct (fully-connected) algorithms"): xGMI connects every pair of GPUs, so
//   1. all-to-all: rank r receives slice r of the range from every rank (every link carries 1 / N of the range, all at once),
//   2. ffh_sum_slices_f32: the N copies added in RANK order -- the same fp32 chain on whichever rank owns the slice: every rank ends up with
//      the same bits, and a run gives the same bits as the next,
//   3. all-gather of the sums (again 1 / N per link), copied back into the range.
// Both collectives go to the buckets' channel where the transport has one.  [ref: one ncclAllReduce per parameter,
// src/runtime/optimizer_kernel.cu:170-171]
int FFModel::allreduce_grads(float* buf, int64_t count, ffh_stream s, bool bucket) const {
  const int G = world_size;
  auto ring = [&]() -> int {
    auto fn = (bucket && config.comm.allreduce_bucket_sum_f32) ? config.comm.allreduce_bucket_sum_f32 : config.comm.allreduce_sum_f32;
    return fn(config.comm.user, buf, count, s);
  };
  auto a2a = config.comm.alltoall_bucket_f32 ? config.comm.alltoall_bucket_f32 : config.comm.alltoall_f32;
  auto gather = config.comm.allgather_bucket_f32 ? config.comm.allgather_bucket_f32 : config.comm.allgather_f32;
  if (!config.direct_allreduce || G < 2 || !gather || !a2a || count <= 0) return ring();
  const int64_t slice = (((count + G - 1) / G) + 3) / 4 * 4;
  if ((size_t)(2 * slice * G) > ar_scratch_floats) return ring();
  // (the count arrays live as long as the model: a transport may key its own bookkeeping on their addresses, as TorchComm does)
  auto& plan = direct_plans[count];
  if (plan.empty()) {
    plan.resize(2 * (size_t)G);
    for (int p = 0; p < G; p++) plan[p] = std::max<int64_t>(0, std::min<int64_t>(slice, count - (int64_t)p * slice));
    for (int q = 0; q < G; q++) plan[G + q] = plan[rank];
  }
  const int64_t* sc = plan.data();
  const int64_t* rc = plan.data() + G;
  const int64_t mine = sc[rank];
  float* r1 = ar_scratch;                    // [G][mine]: slice `rank` as every rank holds it
  float* r2 = ar_scratch + (size_t)slice * G;   // [G][slice]: the sums
  if (a2a(config.comm.user, buf, sc, r1, rc, s) != 0) return 1;
  if (api->ffh_sum_slices_f32(ctx, r1, r1, G, mine, mine, s) != FFH_OK) return 1;
  if (gather(config.comm.user, r1, r2, slice, s) != 0) return 1;
  if (api->ffh_memcpy_d2d(ctx, buf, r2, (size_t)count * 4, s) != FFH_OK) return 1;
  n_direct_allreduces++;
  return 0;
}

bool FFModel::bucketed_now() const {
  if (!exchange || grad_buckets.empty() || use_workers() || config.profiling) return false;
  if (config.bucket_allreduce == 0) return false;
  if (config.bucket_allreduce == 1) return true;
  // by default: where the transport only enqueues (a host-blocking one would stall the launches of the rest of the backward) and the
  // slab makes at least two buckets -- a single one cannot start before the last weight gradient anyway, and its detour over the
  // bucket stream costs two event hops (Kaggle shape, one forced rank: 0.209 vs 0.198 ms)
  return config.comm.nonblocking != 0 && grad_buckets.size() >= 2;
}
int FFModel::big_dw_chunks_now() const {
  if (!bucketed_now()) return 1;
  int n = 0;
  for (const GradBucket& b : grad_buckets) n += b.chunk_layer >= 0;
  return n > 1 ? n : 1;
}
// Issues every bucket whose layers (indices > next_layer) have all issued their backward.  The bucket's stream waits for what the
// compute stream and the weight-gradient streams hold at this point -- the layers' dW / db launches among it -- then runs the sum.
// While a capture is open (--capture-exchange) the sum goes on the capturing stream itself: with RCCL work on a stream that joined
// the capture through an event hipStreamEndCapture recurses (profiles/r04_capture_exchange_endcapture_backtrace.txt).
bool FFModel::buckets_held() const {
  return !config.comm.bucket_channel_own && !shards.empty() && !embeddings.empty() && fused_embedding_update() && !bwd_alltoall_issued;
}
void FFModel::issue_grad_buckets(int next_layer) {
  // A transport that serves the buckets on the SAME channel as the all-to-alls (ffcomm.bucket_channel_own == 0: one RCCL communicator runs its
  // collectives in issue order, whatever streams they are on): nothing goes out before this step's backward all-to-all has been enqueued --
  // otherwise the exchange of the embedding gradients, the table update and the next gather behind it would wait for the biggest layer's
  // weight-gradient GEMM and its all-reduce (round-5 advisor).  The held buckets follow at the next layer boundary.
  if (buckets_held()) return;
  for (size_t k = 0; k < grad_buckets.size(); k++) {
    GradBucket& b = grad_buckets[k];
    if (b.issued || b.lowest_layer <= next_layer) continue;
    issue_one_bucket(k, true);
  }
}
void FFModel::issue_one_bucket(size_t k, bool wait_main) {
  GradBucket& b = grad_buckets[k];
  const bool inline_now = capturing_trace >= 0 || config.capture_exchange;
  ffh_stream s = inline_now ? stream : ar_stream;
  if (dw_worker) dw_worker->drain();
  if (!inline_now && wait_main) {
    check(api->ffh_event_record(ctx, b.ready, stream), "bucket ready");
    check(api->ffh_stream_wait_event(ctx, s, b.ready), "bucket ready");
  }
  // (the weight-gradient streams: joined where this step has used them so far -- a bucket whose layers kept everything on `stream`
  //  waits for nothing extra; inline, `stream` itself takes the waits)
  if (dw_forked && dw1_used) { check(api->ffh_event_record(ctx, b.ready_dw, dw_stream), "bucket ready"); check(api->ffh_stream_wait_event(ctx, s, b.ready_dw), "bucket ready"); }
  if (k < 8) probe_record(14 + 2 * (int)k, s, ctx);
  if (allreduce_grads(mlp_grads + b.off, (int64_t)b.count, s, true) != 0) die("allreduce (bucket) failed");
  if (k < 8) probe_record(15 + 2 * (int)k, s, ctx);
  if (!inline_now) check(api->ffh_event_record(ctx, b.done, s), "bucket done");
  b.issued = true;
  b.inline_issued = inline_now;
  n_bucket_allreduces++;
}

void FFModel::reset_metrics() {
  if (replaying_trace >= 0) return;
  check(api->ffh_zero(ctx, d_perf, sizeof(ffh_perf_metrics), stream), "reset_metrics");
}

// tensor-op mode: the weights' bf16 twin after a host write / (re)initialisation.  Called where a step STARTS -- from begin_trace()
// ahead of a capture or a replay (a replayed forward() returns at once, and a conversion captured into the graph would run on
// every replay) and from forward() for eager steps: the captured GEMMs of a replayed step never read a stale twin.
void FFModel::refresh_weight_twin() const {
  if (!w_twin || !w_twin_dirty) return;
  if (config.fp32_split_bf16x3 && !config.allow_tensor_op_math_conversion) check(api->ffh_convert_f32_to_bf16x3(ctx, mlp_weights, 1, (int64_t)mlp_count, (int64_t)mlp_count, stream), "weight image");
  else check(api->ffh_convert_f32_to_bf16(ctx, w_twin, mlp_weights, (int64_t)mlp_count, stream), "weight twin");
  w_twin_dirty = false;
}
void FFModel::note_weight_write(const void* p) const {
  if (w_twin && (const char*)p >= (const char*)mlp_weights && (const char*)p < (const char*)(mlp_weights + mlp_count)) w_twin_dirty = true;
}

void FFModel::forward(int _seq_length) {
  if (replaying_trace >= 0) return;
  seq_length = _seq_length;
  if (capturing_trace < 0) refresh_weight_twin();      // (a capture: begin_trace() did it on the stream, outside the graph)
  emb_forward_issued = emb_forward_joined = false;
  // gather (+ all-to-all) go to the side stream beside the bottom MLP: the fork point is here (inputs ready)
  if (config.overlap_embedding && !embeddings.empty()) {
    // The side stream already runs behind everything it depends on from earlier steps (the table update is on it);
    // what it must additionally see is a batch that was copied in on `stream`.  No new batch (the reference reuses
    // the warm-up batch for random input), no event: each record / wait is a barrier packet on the critical stream.
    // Exception: data-parallel (replicated) tables live in the dense parameter slab, which the optimizer of the step before
    // wrote on `stream` (all-reduce + SGD / Adam in update()): their gather must always be ordered behind it.
    // ... and so must the gather of EVERY table when the tables are updated by the dense path, which runs on `stream` in update().
    fork_recorded = inputs_dirty || capturing_trace >= 0 || use_workers() || repl_workspace != nullptr || !fused_embedding_update();
    if (fork_recorded) check(api->ffh_event_record(ctx, ev_fork, stream), "fork");
    inputs_dirty = false;
    // start the gather right now unless a host-side collective would stall THIS thread's launches
    if (!exchange || config.comm.nonblocking || use_workers()) issue_embedding_forward_on_side_stream();
  }
  for (Op* op : layers) {
    if (!config.profiling) { op->forward(*this); continue; }
    if (op->op_type == OP_EMBEDDING && emb_forward_issued) continue;      // the first table launched the whole group
    profiled(op, true, [&] { op->forward(*this); });
  }
  if (emb_forward_issued && !emb_forward_joined) join_embedding_forward();
}

// One op between two events on `stream`, waited for and printed in the reference's formats
// [ref: src/ops/linear.cu:525-546,761; src/ops/concat.cu:282-297,400-412; src/ops/batch_matmul.cu:303-318,476-494].
void FFModel::profiled(const Op* op, bool fwd, const std::function<void()>& fn) const {
  ffh_event e0, e1;
  check(api->ffh_event_create(ctx, &e0), "event");
  check(api->ffh_event_create(ctx, &e1), "event");
  check(api->ffh_event_record(ctx, e0, stream), "event");
  fn();
  check(api->ffh_event_record(ctx, e1, stream), "event");
  check(api->ffh_event_sync(ctx, e1), "event");
  float ms = 0.0f;
  check(api->ffh_event_elapsed_ms(ctx, e0, e1, &ms), "event");
  api->ffh_event_destroy(ctx, e0);
  api->ffh_event_destroy(ctx, e1);
  const char* dir = fwd ? "forward" : "backward";
  switch (op->op_type) {
    case OP_LINEAR:
      if (fwd) printf("%s [Linear] forward time = %.2lfms\n", op->name, (double)ms);
      else printf("Linear backward time = %.2lfms\n", (double)ms);
      break;
    case OP_BATCHMATMUL: printf("BatchMatmul %s time = %.2lfms\n", dir, (double)ms); break;
    case OP_EMBEDDING:   // the reference dumps tensors here (src/ops/embedding.cu:266-271); one launch serves every table
      printf("[Embedding x%zu] %s time = %.4f ms\n", embeddings.size(), dir, ms);
      break;
    default: printf("[%s] %s time = %.4f ms\n", op->name, dir, ms); break;   // Concat's format (its backward also says "forward" in the reference, :412)
  }
}

void FFModel::issue_embedding_forward_on_side_stream() const {
  if (use_workers()) {
    const FFModel* self = this;
    side_worker->post([self](ffh_ctx* wc) {
      self->check(self->api->ffh_stream_wait_event(wc, self->side_stream, self->ev_fork), "fork");
      self->embedding_group_forward(self->side_stream, wc);
      self->check(self->api->ffh_event_record(wc, self->ev_join, self->side_stream), "join");
    });
  } else {
    if (fork_recorded) check(api->ffh_stream_wait_event(ctx, side_stream, ev_fork), "fork");
    probe_record(0, side_stream, ctx);
    embedding_group_forward(side_stream);
    probe_record(1, side_stream, ctx);
    check(api->ffh_event_record(ctx, ev_join, side_stream), "join");
    if (early_sort_possible(1)) {       // behind the join: nothing waits for it until this step's update
      launch_shard_groups(this, kSortOnly, side_stream, ctx);
      emb_sorted_early = true;
    }
  }
  emb_forward_issued = true;
  emb_forward_joined = false;
}

void FFModel::join_embedding_forward() const {
  if (side_worker) side_worker->drain();   // the record of ev_join must have been issued before we wait on it
  probe_record(10, stream, ctx);           // bench probes: what the compute stream waits here is the EXPOSED part of gather + exchange
  check(api->ffh_stream_wait_event(ctx, stream, ev_join), "join");
  probe_record(11, stream, ctx);
  emb_forward_joined = true;
}

void FFModel::issue_embedding_update_on_side_stream() const {
  if (use_workers()) {
    const FFModel* self = this;
    side_worker->post([self](ffh_ctx* wc) {
      self->check(self->api->ffh_stream_wait_event(wc, self->side_stream, self->ev_grad_ready), "event");
      self->embedding_group_update(self->side_stream, wc);
      self->check(self->api->ffh_event_record(wc, self->ev_update_done, self->side_stream), "event");
    });
  } else {
    check(api->ffh_stream_wait_event(ctx, side_stream, ev_grad_ready), "event");
    probe_record(2, side_stream, ctx);
    embedding_group_update(side_stream);
    probe_record(3, side_stream, ctx);
    check(api->ffh_event_record(ctx, ev_update_done, side_stream), "event");
  }
}

// Writers of the model inputs on `stream` (the data loader's next batch) go behind the side-stream table update of the
// step before, which still sorts and reads the sparse ids.
void FFModel::order_input_writes_behind_update() const {
  if (embeddings.empty() || !config.overlap_embedding || !fused_embedding_update()) return;
  if (side_worker) side_worker->drain();       // the record of ev_update_done must have been issued
  check(api->ffh_stream_wait_event(ctx, stream, ev_update_done), "inputs behind the table update");
}

void FFModel::zero_gradients() {
  if (replaying_trace >= 0) return;
  // Op::zero_grad for every layer [ref: src/runtime/model.cc:466-490]: two slabs instead of ~34 tasks.
  // Embedding tables have no dense gradient on the fused path (nothing to zero: SURVEY fact 1).
  // activation gradients with a single producer are stored, not accumulated: nothing to clear (0 + x == x)
  if (need_zero_act_grads) check(api->ffh_zero(ctx, act_grad_slab, act_grad_bytes, stream), "zero_gradients");
  if (!mlp_grads_clean) check(api->ffh_zero(ctx, mlp_grads, mlp_count * 4, stream), "zero_gradients");
  if (exchange && gsend && need_zero_gsend) {
    size_t n = 0;
    for (int64_t c : fwd_recv_counts) n += (size_t)c;
    check(api->ffh_zero(ctx, gsend, n * 4, stream), "zero_gradients");
  }
  if (!fused_embedding_update())
    for (Embedding* e : embeddings)
      if (e->held_here(rank) && !e->replicated)
        check(api->ffh_zero(ctx, e->weights[0].impl->grad, e->weights[0].impl->bytes + (e->row_sharded ? (size_t)e->out_channels * 4 : 0), stream), "zero_gradients");
}

void FFModel::compute_metrics() {
  if (replaying_trace >= 0) return;
  const Tensor& fin = layers.back()->outputs[0];
  check(api->ffh_metrics_update(ctx, (const float*)fin.impl->ptr, (const float*)label_tensor.impl->ptr, d_perf,
                                local_rows(fin, this), fin.adim[0], metrics_flags, stream), "compute_metrics");
}

void FFModel::backward(int _seq_length) {
  if (replaying_trace >= 0) return;
  seq_length = _seq_length;
  if (config.computationMode != COMP_MODE_TRAINING) die("backward() in inference mode");
  // compute_metrics() + loss backward [ref: src/runtime/model.cc:1443-1452; src/loss_functions/loss_functions.cu:141-170,196-237]
  // in one launch; scale_factor = 1 / global batch
  const Tensor& fin = layers.back()->outputs[0];
  const float scale = loss_type == LOSS_MEAN_SQUARED_ERROR_AVG_REDUCE ? 1.0f / (float)fin.adim[fin.numDim - 1] : 1.0f;
  if (fin.impl->grad_ld != fin.adim[0] || fin.impl->ld != fin.adim[0]) die("final layer output must be contiguous");
  dw_forked = false;
  opt_next_done = false;
  // the click-probability layer (out = 1): loss step + metrics + the layer's whole backward in ONE launch; any other last
  // layer: the loss kernel, then the layer's own backward
  int first = (int)layers.size() - 1;
  emb_update_pending = false;
  mlp_grads_clean = false;
  Linear* last = (config.fuse_loss && !config.profiling) ? dynamic_cast<Linear*>(layers.back()) : nullptr;   // --profiling: the loss kernel and every layer on their own
  int rc = FFH_ERR_UNSUPPORTED;
  if (last) {
    const Tensor& x = last->inputs[0];
    const int flags = (last->dx_overwrite ? FFH_LINEAR_DX_OVERWRITE : 0) | (last->dx_mask_by_x ? FFH_LINEAR_DX_MASK_BY_X : 0);
    rc = api->ffh_linear_bwd_mse(ctx, (const float*)x.impl->ptr, x.impl->ld, last->discard_input_grad ? nullptr : x.impl->grad, x.impl->grad_ld,
                                 (const float*)fin.impl->ptr, fin.impl->ld, fin.impl->grad, fin.impl->grad_ld,
                                 (const float*)last->weights[0].impl->ptr, last->weights[0].impl->grad,
                                 last->use_bias ? last->weights[1].impl->grad : nullptr, last->in_channels, last->out_channels,
                                 local_rows(fin, this), (int)last->activation, flags, (const float*)label_tensor.impl->ptr, scale, d_perf,
                                 metrics_flags, stream);
    if (rc == FFH_OK) first--;                          // the last layer is done
    else if (rc != FFH_ERR_UNSUPPORTED) check(rc, "loss + last layer backward");
  }
  if (rc != FFH_OK)
    check(api->ffh_mse_bwd_metrics(ctx, fin.impl->grad, (const float*)fin.impl->ptr, (const float*)label_tensor.impl->ptr, d_perf,
                                   local_rows(fin, this), fin.adim[0], scale, metrics_flags, stream), "metrics + loss backward");
  grad_ready_attached = false;
  z_free_recorded = false;
  auto mark_z_free = [&](int l) {     // behind the last reader of the gather's destination among the forked weight-gradient GEMMs
    if (l == z_reader_layer && dw_forked && !dw_worker && capturing_trace < 0) {
      check(api->ffh_event_record(ctx, ev_z_free, dw_stream), "z free");
      z_free_recorded = true;
    }
  };
  bwd_alltoall_issued = false;
  for (GradBucket& b : grad_buckets) b.issued = b.inline_issued = false;
  const int dw_chunks = big_dw_chunks_now();
  // the biggest layer with the bucketed all-reduce: data gradient, then its weight gradient in row blocks, a bucket behind each
  auto chunked_big_backward = [&](Linear* up, int l) {
    // (dy is final here: the weight-gradient stream forks in FRONT of the data gradient, as the library's own fork does, and the GEMMs of
    //  the row blocks run beside it)
    ffh_event ev = layer_events[l];
    check(api->ffh_event_record(ctx, ev, stream), "event");
    check(api->ffh_stream_wait_event(ctx, dw_stream, ev), "event");
    up->backward_part(*this, 1);
    const int per = up->out_channels / dw_chunks;
    for (size_t k = 0; k < grad_buckets.size(); k++) {
      GradBucket& b = grad_buckets[k];
      if (b.chunk_layer != l) continue;
      up->backward_dw_rows(*this, b.chunk_index * per, per);
      if (!buckets_held()) issue_one_bucket(k, false);      // (held: issue_grad_buckets sends it once the backward all-to-all is enqueued)
    }
    up->db_from_upper = false;
  };
  for (int l = first; l >= 0; l--) {
    if (bucketed_now()) issue_grad_buckets(l);     // the buckets every layer above l has completed
    if (l == grad_attach_layer) {
      check(api->ffh_event_record_with_next_linear_bwd(ctx, ev_grad_ready), "attach event");
      grad_ready_attached = true;
    }
    Linear* up = layers[l]->op_type == OP_LINEAR ? static_cast<Linear*>(layers[l]) : nullptr;
    if (up && up->dx_map && !use_workers()) {
      const bool attach = l == scatter_attach_layer;
      check(api->ffh_linear_bwd_set_dx_scatter(ctx, up->dx_map, up->in_channels, attach ? ev_grad_ready : nullptr), "dx scatter");
      if (dw_chunks > 1 && l == big_dw_layer) { chunked_big_backward(up, l); mark_z_free(l); }
      else { up->backward(*this); mark_z_free(l); }
      if (api->ffh_linear_dx_scatter_used(ctx)) {
        up->dx_map_concat->bwd_done = true;                    // its pack kernel is not needed this step
        if (attach) grad_ready_attached = true;
      }
      continue;
    }
    if (config.profiling) {
      if (layers[l]->op_type == OP_EMBEDDING && static_cast<Embedding*>(layers[l])->table_index != (int)embeddings.size() - 1) continue;
      profiled(layers[l], false, [&] {
        layers[l]->backward(*this);
        // the fused sparse update of the tables is the embedding group's backward here (it runs in update() otherwise)
        if (layers[l]->op_type == OP_EMBEDDING && fused_embedding_update()) embedding_group_update(stream);
      });
      continue;
    }
    if (up && !up->chain_bwd.empty() && mlp_chain_usable(local_rows(up->outputs[0], this), false)) {
      // the chain this layer tops: one call for all its members (their indices are l - n + 1 .. l)
      const int n = (int)up->chain_bwd.size();
      const int crc = run_chain_bwd(up);
      if (crc == FFH_OK) {
        // a lower member completes the embedding output gradients: "gradients ready" behind the whole call (the chain's weight-gradient
        // kernel still reads the buffer the next gather overwrites).  (l itself: attached above, recorded by the call.)
        if (grad_attach_layer > l - n && grad_attach_layer < l && !grad_ready_attached) {
          check(api->ffh_event_record(ctx, ev_grad_ready, stream), "event");
          grad_ready_attached = true;
        }
        for (int k = 0; k < n; k++) mark_z_free(l - k);
        l -= n - 1;
        continue;
      }
      if (crc != FFH_ERR_UNSUPPORTED) check(crc, up->name);
      up->chain_bwd.clear();                                   // not a chain the library serves: the per-layer calls from now on
    }
    if (up && up->pair_lower && !use_workers() && l != grad_attach_layer && l >= 1 && layers[l - 1] == up->pair_lower) {
      const int prc = up->backward_pair(*this);
      if (prc == FFH_OK) { l--; continue; }                   // the lower layer is done as well
      if (prc != FFH_ERR_UNSUPPORTED) check(prc, up->name);
      up->pair_lower = nullptr;                                // not a shape the pair launch serves: the ordinary calls from now on
    }
    if (up && dw_chunks > 1 && l == big_dw_layer) { chunked_big_backward(up, l); mark_z_free(l); continue; }
    layers[l]->backward(*this);
    mark_z_free(l);
  }
  if (emb_update_pending) {
    // exchange of the row gradients + fused sparse update on the side stream, beside the bottom-MLP backward
    issue_embedding_update_on_side_stream();
    emb_update_pending = false;
  }
  // what is left (a shared channel: the buckets held until the exchange above was enqueued; a model whose backward all-to-all runs in update():
  // released here all the same -- update() waits for every bucket)
  if (bucketed_now()) { bwd_alltoall_issued = true; issue_grad_buckets(-1); }
}

void FFModel::update() {
  if (replaying_trace >= 0) return;
  optimizer->next();
  opt_next_done = true;
  SGDOptimizer* sgd = dynamic_cast<SGDOptimizer*>(optimizer);
  AdamOptimizer* adam = dynamic_cast<AdamOptimizer*>(optimizer);
  if (!sgd && !adam) die("update(): unknown optimizer");
  // every rank must issue its collectives in the same order: the side thread's all-to-all (backward) first
  if (side_worker) side_worker->drain();
  if (dw_forked && !dw_worker && !api->ffh_second_stream_used(ctx, 1) && !dw_stream_used_directly) { dw_forked = false; dw1_used = false; }   // the library kept everything on `stream`
  dw_stream_used_directly = false;
  if (dw_forked) {   // the weight-gradient GEMMs ran on their own stream: join before the gradients are consumed
    if (dw_worker) { dw_worker->drain(); dw1_used = true; }
    if (dw1_used) {
      check(api->ffh_event_record(ctx, ev_dw_done, dw_stream), "join dw");
      check(api->ffh_stream_wait_event(ctx, stream, ev_dw_done), "join dw");
    }
    // the next gather (side stream) overwrites embedding outputs that alias the Concat output -- the x operand of the first
    // top-MLP layer, which a forked dW GEMM may still be reading: write-after-read across streams
    if (config.overlap_embedding && !embeddings.empty() && !use_workers()) {
      if (z_free_recorded) check(api->ffh_stream_wait_event(ctx, side_stream, ev_z_free), "join dw (embedding stream)");
      else {
        if (dw1_used) check(api->ffh_stream_wait_event(ctx, side_stream, ev_dw_done), "join dw (embedding stream)");
      }
    }
    dw1_used = false;
    dw_forked = false;
  }
  // data-parallel MLP gradients: ONE bucket [ref: one ncclAllReduce per tensor, src/runtime/optimizer_kernel.cu:170-171].
  // No 1/world_size: the loss already divides by the global batch.
  if (exchange && mlp_count) {
    bool any_issued = false;
    for (const GradBucket& b : grad_buckets) any_issued = any_issued || b.issued;
    if (any_issued) {
      // the buckets went out from backward() on ar_stream: the optimizer waits for them here (what `stream` stands at these waits is
      // the EXPOSED part of the all-reduce: probe pair 12 / 13), then what no bucket covers is reduced as before
      for (const GradBucket& b : grad_buckets) if (!b.issued) die("update(): a gradient bucket was not issued");
      probe_record(12, stream, ctx);
      for (const GradBucket& b : grad_buckets) if (!b.inline_issued) check(api->ffh_stream_wait_event(ctx, stream, b.done), "join all-reduce bucket");
      probe_record(13, stream, ctx);
      probe_record(8, stream, ctx);
      for (auto& r : grad_rest)
        if (allreduce_grads(mlp_grads + r.first, (int64_t)r.second, stream, false) != 0) die("allreduce failed");
      probe_record(9, stream, ctx);
    } else {
      probe_record(8, stream, ctx);
      if (allreduce_grads(mlp_grads, (int64_t)mlp_count, stream, false) != 0) die("allreduce failed");
      probe_record(9, stream, ctx);
    }
  }
  // one launch over the whole MLP slab; it also clears the gradients it consumed, so the next zero_gradients()
  // has nothing to sweep [ref: one update task per parameter, src/runtime/optimizer.cc:93-189,256-330]
  const size_t opt_count = mlp_count;
  if (adam) {
    if (mlp_count) {
      check(api->ffh_adam_update(ctx, mlp_weights, mlp_grads, adam->mlp_m, adam->mlp_v, (int64_t)opt_count, (float)adam->alpha_t,
                                 (float)adam->beta1, (float)adam->beta2, (float)adam->weight_decay, (float)adam->epsilon,
                                 FFH_OPT_ZERO_GRAD, stream), "adam_update (MLP slab)");
      mlp_grads_clean = true;
    }
  } else if (sgd->momentum > 0.0) {
    for (const Parameter& p : parameters)
      if (in_dense_slab(p)) sgd->update(&p);
  } else if (mlp_count) {
    check(api->ffh_sgd_update_ex(ctx, mlp_weights, mlp_grads, nullptr, (int64_t)opt_count, (float)sgd->lr, (float)sgd->weight_decay, 0.0f,
                                 0, FFH_OPT_ZERO_GRAD, stream), "sgd_update (MLP slab)");
    mlp_grads_clean = true;
  }
  if (fused_embedding_update()) {
    if (config.overlap_embedding) {
      // launched in backward() on the side stream.  Its only consumer, the next gather, runs on that same stream, and
      // every host read of a table syncs both streams -- so `stream` joins it only where a capture must close the fork
      if (!embeddings.empty() && (capturing_trace >= 0 || use_workers()))
        check(api->ffh_stream_wait_event(ctx, stream, ev_update_done), "join update");
    } else if (!config.profiling) {     // (--profiling: timed as the embedding group's backward)
      embedding_group_update(stream);
    }
  } else {
    embedding_dense_update();
  }
}

bool FFModel::trace_replays(int trace_id) const {
  if (!config.enable_graph) return false;
  auto it = trace_tune.find(trace_id);
  return !(trace_adaptive() && it != trace_tune.end() && it->second.decided == 2);
}

// Adaptive replay (FFConfig::trace_mode 0, one GPU): calls 0-4 of a trace run eagerly -- the first two unmeasured (one-off costs of a first
// launch: code-object loads, hipFuncSetAttribute, a fall-back path taken once; a cold sample biased the choice towards the replay: round-5
// advisor), events behind calls 2 and 4 --, call 5 captures, calls 6-8 replay with events behind 6 and 8; the tenth call compares the spacing
// of the events (eager steps 3-4 against replays 7-8: what a step takes end to end, host gaps included) and keeps the faster form for good.
void FFModel::begin_trace(int trace_id) {
  if (!config.enable_graph) return;
  if (trace_adaptive()) {
    TraceTune& t = trace_tune[trace_id];
    if (t.decided == 2 || (t.decided == 0 && t.calls < 5)) return;      // an eager step
    if (t.decided == 0 && t.calls == 9) {
      check(api->ffh_event_sync(ctx, t.ev[3]), "trace timing");
      check(api->ffh_event_elapsed_ms(ctx, t.ev[0], t.ev[1], &t.eager_ms), "trace timing");
      check(api->ffh_event_elapsed_ms(ctx, t.ev[2], t.ev[3], &t.graph_ms), "trace timing");
      t.decided = t.graph_ms <= 1.02f * t.eager_ms ? 1 : 2;
      for (ffh_event& e : t.ev) { api->ffh_event_destroy(ctx, e); e = nullptr; }
      if (config.profiling || getenv("FFM_TRACE_VERBOSE"))
        fprintf(stderr, "[DLRM] trace %d: eager %.1f us / step, hipGraph replay %.1f us / step -> %s\n", trace_id, t.eager_ms * 500.f, t.graph_ms * 500.f,
                t.decided == 1 ? "replay" : "eager");
      if (t.decided == 2) return;
    }
  }
  refresh_weight_twin();                  // ahead of the capture / the replay, on `stream`
  auto it = graphs.find(trace_id);
  if (it != graphs.end()) { replaying_trace = trace_id; return; }
  if (dw_worker) dw_worker->drain();      // stream capture is thread-local: everything is issued inline while capturing
  if (side_worker) side_worker->drain();
  int rc = api->ffh_graph_begin_capture(ctx, stream);
  if (rc == FFH_ERR_UNSUPPORTED) { config.enable_graph = false; return; }   // backend without graphs: run eagerly
  check(rc, "begin_trace");
  capturing_trace = trace_id;
}

void FFModel::end_trace(int trace_id) {
  if (!config.enable_graph) return;
  TraceTune* tune = nullptr;
  if (trace_adaptive()) {
    TraceTune& t = trace_tune[trace_id];
    if (t.decided == 2) return;
    if (t.decided == 0) {
      auto mark = [&](int k) {
        if (!t.ev[k]) check(api->ffh_event_create(ctx, &t.ev[k]), "event create");
        check(api->ffh_event_record(ctx, t.ev[k], stream), "trace timing");
      };
      if (t.calls < 5) {               // the eager steps: events behind the third and the fifth
        if (t.calls == 2) mark(0);
        if (t.calls == 4) mark(1);
        t.calls++;
        return;
      }
      tune = &t;
    }
  }
  struct TuneMark {                    // behind the graph launch below (calls 6 and 8: two replays apart)
    FFModel* ff; TraceTune* t;
    ~TuneMark() {
      if (!t) return;
      const int k = t->calls == 6 ? 2 : (t->calls == 8 ? 3 : -1);
      if (k >= 0) {
        if (!t->ev[k]) ff->check(ff->api->ffh_event_create(ff->ctx, &t->ev[k]), "event create");
        ff->check(ff->api->ffh_event_record(ff->ctx, t->ev[k], ff->stream), "trace timing");
      }
      t->calls++;
    }
  } tune_mark{this, tune};
  if (capturing_trace == trace_id) {
    ffh_graph g = nullptr;
    check(api->ffh_graph_end_capture(ctx, stream, &g), "end_trace");
    graphs[trace_id] = g;
    capturing_trace = -1;
    check(api->ffh_graph_launch(ctx, g, stream), "graph launch");   // the captured iteration has not run yet
    return;
  }
  if (replaying_trace == trace_id) {
    check(api->ffh_graph_launch(ctx, graphs[trace_id], stream), "graph launch");
    replaying_trace = -1;
  }
}

void FFModel::sync() {
  if (dw_worker) dw_worker->drain();
  if (side_worker) side_worker->drain();
  check(api->ffh_stream_sync(ctx, stream), "sync");
  check(api->ffh_stream_sync(ctx, side_stream), "sync");
  check(api->ffh_stream_sync(ctx, dw_stream), "sync");
  check(api->ffh_stream_sync(ctx, ar_stream), "sync");
}

PerfMetrics FFModel::get_perf_metrics() {
  sync();
  ffh_perf_metrics h;
  check(api->ffh_memcpy_d2h(ctx, &h, d_perf, sizeof h, stream), "metrics d2h");
  check(api->ffh_stream_sync(ctx, stream), "sync");
  PerfMetrics p;
  p.train_all = h.train_all; p.train_correct = h.train_correct; p.cce_loss = h.cce_loss;
  p.sparse_cce_loss = h.sparse_cce_loss; p.mse_loss = h.mse_loss; p.rmse_loss = h.rmse_loss; p.mae_loss = h.mae_loss;
  return p;
}
