// model.cc -- FFModel / Op / Optimizer / Initializer of the reference API over the kernel C-ABI.
// See ffmodel.h for the reference declarations each class mirrors.
#include "ffmodel.h"
#include "model_internal.h"


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
  // tensor-op mode, bwd_exact: the twin of the data gradient the exact kernels stored
  auto image_dx = [&] {
    if (dx_image && dx) ff.check(ff.api->ffh_convert_f32_to_bf16x3(ff.ctx, dx, b, in_channels, lddx, ff.stream), name);
    if (bwd_exact && dx && dx_twin) ff.check(ff.api->ffh_convert_f32_to_bf16(ff.ctx, dx_twin, dx, b * (int64_t)in_channels, ff.stream), name);
  };
  struct ExactScope {      // (allocate() step 7: bwd_exact)
    const FFModel& ff; bool on;
    ExactScope(const FFModel& f, bool o) : ff(f), on(o) { if (on) ff.check(ff.api->ffh_ctx_set_math_mode(ff.ctx, FFH_MATH_DEFAULT), "math mode"); }
    ~ExactScope() { if (on) ff.check(ff.api->ffh_ctx_set_math_mode(ff.ctx, FFH_MATH_TENSOR_OP_BF16), "math mode"); }
  } exact_scope(ff, bwd_exact && !ff.use_workers());
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
  // (the split mode leaves the narrow layers of a chain to the exact kernels: the library takes the chain there, allocate() step 7 drops a chain
  //  with a member that keeps an image by conversion)
  return config.mlp_chain && !config.profiling && !use_workers() && !config.allow_tensor_op_math_conversion && rows <= config.mlp_chain_max_batch && (!fwd || (rows >= config.mlp_chain_fwd_min_batch && rows <= config.mlp_chain_fwd_max_batch));
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
