// ffmodel_c.cc -- see ffmodel_c.h
#include "ffmodel_c.h"
#include <string>

#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "backend.h"
#include "dlrm.h"
#include "ffmodel.h"

namespace {
FFConfig* C(flexflow_config_t h) { return (FFConfig*)h.impl; }
FFModel* M(flexflow_model_t h) { return (FFModel*)h.impl; }
Tensor* T(flexflow_tensor_t h) { return (Tensor*)h.impl; }
Initializer* I(flexflow_initializer_t h) { return (Initializer*)h.impl; }
DLRMApp* A(flexflow_dlrm_t h) { return (DLRMApp*)h.impl; }
flexflow_tensor_t wrap(const Tensor& t) { flexflow_tensor_t h; h.impl = new Tensor(t); return h; }   // handles live as long as the process
std::vector<int> dims_vec(const int* dims, int n) { return std::vector<int>(dims, dims + n); }
}  // namespace

extern "C" {

flexflow_config_t flexflow_config_create(void) { flexflow_config_t h; h.impl = new FFConfig(); return h; }
void flexflow_config_destroy(flexflow_config_t h) { delete C(h); }
void flexflow_config_parse_args(flexflow_config_t h, char** argv, int argc) { C(h)->parse_args(argv, argc); }
void flexflow_config_set_comm(flexflow_config_t h, const ffcomm* comm) { if (comm) C(h)->comm = *comm; }
void flexflow_config_set_batch_size(flexflow_config_t h, int b) { C(h)->batchSize = b; }
int  flexflow_config_get_batch_size(flexflow_config_t h) { return C(h)->batchSize; }
void flexflow_config_set_backend(flexflow_config_t h, const char* p) { C(h)->backend_lib = p ? p : ""; }
void flexflow_config_set_seed(flexflow_config_t h, uint64_t s) { C(h)->seed = s; }
void flexflow_config_set_device(flexflow_config_t h, int d) { C(h)->device = d; }
void flexflow_config_set_enable_graph(flexflow_config_t h, bool v) { C(h)->enable_graph = v; }
void flexflow_config_set_overlap_embedding(flexflow_config_t h, bool v) { C(h)->overlap_embedding = v; }
void flexflow_config_set_dense_embedding_update(flexflow_config_t h, bool v) { C(h)->dense_embedding_update = v; }

flexflow_model_t flexflow_model_create(flexflow_config_t c) { flexflow_model_t h; h.impl = new FFModel(*C(c)); return h; }
void flexflow_model_destroy(flexflow_model_t h) { delete M(h); }

flexflow_tensor_t flexflow_tensor_create(flexflow_model_t m, int nd, const int* dims, int dt, bool cg) {
  switch (nd) {
    case 1: return wrap(M(m)->create_tensor<1>(dims, (DataType)dt, NULL, cg));
    case 2: return wrap(M(m)->create_tensor<2>(dims, (DataType)dt, NULL, cg));
    case 3: return wrap(M(m)->create_tensor<3>(dims, (DataType)dt, NULL, cg));
    case 4: return wrap(M(m)->create_tensor<4>(dims, (DataType)dt, NULL, cg));
    default: fprintf(stderr, "FATAL: tensors have 1..4 dimensions\n"); abort();
  }
}
flexflow_tensor_t flexflow_model_add_dense(flexflow_model_t m, flexflow_tensor_t in, int out_dim, int act, bool use_bias,
                                           flexflow_initializer_t ki, flexflow_initializer_t bi, const char* name) {
  return wrap(M(m)->dense(*T(in), out_dim, (ActiMode)act, use_bias, NULL, I(ki), I(bi), name));
}
flexflow_tensor_t flexflow_model_add_embedding(flexflow_model_t m, flexflow_tensor_t in, int num_entries, int out_dim, int aggr,
                                               flexflow_initializer_t ki, const char* name) {
  return wrap(M(m)->embedding(*T(in), num_entries, out_dim, (AggrMode)aggr, NULL, I(ki), name));
}
flexflow_tensor_t flexflow_model_add_concat(flexflow_model_t m, int n, const flexflow_tensor_t* ins, int axis, const char* name) {
  std::vector<Tensor> v;
  for (int i = 0; i < n; i++) v.push_back(*T(ins[i]));
  return wrap(M(m)->concat(n, v.data(), axis, name));
}
flexflow_tensor_t flexflow_model_add_flat(flexflow_model_t m, flexflow_tensor_t in, const char* name) { return wrap(M(m)->flat(*T(in), name)); }
flexflow_tensor_t flexflow_model_add_dot_interaction(flexflow_model_t m, flexflow_tensor_t in, int d, const char* name) { return wrap(M(m)->dot_interaction(*T(in), d, name)); }
flexflow_tensor_t flexflow_model_add_tril(flexflow_model_t m, flexflow_tensor_t in, const char* name) { return wrap(M(m)->tril(*T(in), name)); }
flexflow_tensor_t flexflow_model_add_transpose(flexflow_model_t m, flexflow_tensor_t in, int n, const int* perm, const char* name) {
  return wrap(M(m)->transpose(*T(in), dims_vec(perm, n), name));
}
flexflow_tensor_t flexflow_model_add_reshape(flexflow_model_t m, flexflow_tensor_t in, int n, const int* shape, const char* name) {
  return wrap(M(m)->reshape(*T(in), dims_vec(shape, n), name));
}
flexflow_tensor_t flexflow_model_add_batch_matmul(flexflow_model_t m, flexflow_tensor_t a, flexflow_tensor_t b, int asd, int bsd) {
  return wrap(M(m)->batch_matmul(*T(a), *T(b), asd, bsd));
}
flexflow_initializer_t flexflow_zero_initializer_create(void) { flexflow_initializer_t h; h.impl = new ZeroInitializer(); return h; }
flexflow_initializer_t flexflow_uniform_initializer_create(int seed, float lo, float hi) { flexflow_initializer_t h; h.impl = new UniformInitializer(seed, lo, hi); return h; }
flexflow_initializer_t flexflow_norm_initializer_create(int seed, float mean, float sd) { flexflow_initializer_t h; h.impl = new NormInitializer(seed, mean, sd); return h; }
flexflow_initializer_t flexflow_glorot_uniform_initializer_create(int seed) { flexflow_initializer_t h; h.impl = new GlorotUniform(seed); return h; }
flexflow_sgd_optimizer_t flexflow_sgd_optimizer_create(flexflow_model_t m, double lr, double mom, bool nest, double wd) {
  flexflow_sgd_optimizer_t h; h.impl = new SGDOptimizer(M(m), lr, mom, nest, wd); return h;
}
void flexflow_model_set_sgd_optimizer(flexflow_model_t m, flexflow_sgd_optimizer_t o) { M(m)->optimizer = (SGDOptimizer*)o.impl; }
flexflow_adam_optimizer_t flexflow_adam_optimizer_create(flexflow_model_t m, double alpha, double b1, double b2, double wd, double eps) {
  flexflow_adam_optimizer_t h; h.impl = new AdamOptimizer(M(m), alpha, b1, b2, wd, eps); return h;
}
void flexflow_adam_optimizer_set_lr(flexflow_adam_optimizer_t o, double lr) { ((AdamOptimizer*)o.impl)->alpha = lr; }
void flexflow_model_set_adam_optimizer(flexflow_model_t m, flexflow_adam_optimizer_t o) { M(m)->optimizer = (AdamOptimizer*)o.impl; }
void flexflow_model_compile(flexflow_model_t m, int loss, const int* metrics, int nb, int comp_mode) {
  std::vector<MetricsType> v;
  for (int i = 0; i < nb; i++) v.push_back((MetricsType)metrics[i]);
  M(m)->compile((LossType)loss, v, (CompMode)comp_mode);
}
void flexflow_model_init_layers(flexflow_model_t m) { M(m)->init_layers(); }
void flexflow_model_reset_metrics(flexflow_model_t m) { M(m)->reset_metrics(); }
void flexflow_model_forward(flexflow_model_t m, int sl) { M(m)->forward(sl); }
void flexflow_model_zero_gradients(flexflow_model_t m) { M(m)->zero_gradients(); }
void flexflow_model_backward(flexflow_model_t m, int sl) { M(m)->backward(sl); }
void flexflow_model_update(flexflow_model_t m) { M(m)->update(); }
void flexflow_model_begin_trace(flexflow_model_t m, int id) { M(m)->begin_trace(id); }
void flexflow_model_end_trace(flexflow_model_t m, int id) { M(m)->end_trace(id); }
void flexflow_model_sync(flexflow_model_t m) { M(m)->sync(); }
const char* flexflow_model_get_backend_name(flexflow_model_t m) { return M(m)->api->ffh_backend_name(); }
const char* flexflow_model_get_backend_path(flexflow_model_t m) { return M(m)->api->path.c_str(); }
void flexflow_model_get_perf_metrics(flexflow_model_t m, flexflow_perf_metrics_t* out) {
  PerfMetrics p = M(m)->get_perf_metrics();
  out->train_all = p.train_all; out->train_correct = p.train_correct; out->cce_loss = p.cce_loss;
  out->sparse_cce_loss = p.sparse_cce_loss; out->mse_loss = p.mse_loss; out->rmse_loss = p.rmse_loss; out->mae_loss = p.mae_loss;
}
flexflow_tensor_t flexflow_model_get_label_tensor(flexflow_model_t m) { return wrap(M(m)->label_tensor); }
int flexflow_model_get_num_layers(flexflow_model_t m) { return (int)M(m)->layers.size(); }
const char* flexflow_model_get_layer_name(flexflow_model_t m, int l) { return M(m)->layers.at(l)->name; }
int flexflow_model_get_layer_num_weights(flexflow_model_t m, int l) { return M(m)->layers.at(l)->numWeights; }
flexflow_tensor_t flexflow_model_get_parameter(flexflow_model_t m, int l, int i) {
  Op* op = M(m)->layers.at(l);
  if (i < 0 || i >= op->numWeights) { fprintf(stderr, "FATAL: %s has %d weights\n", op->name, op->numWeights); abort(); }
  return wrap(op->weights[i]);
}
flexflow_tensor_t flexflow_model_get_layer_output(flexflow_model_t m, int l) { return wrap(M(m)->layers.at(l)->outputs[0]); }
void* flexflow_model_get_stream(flexflow_model_t m) { return M(m)->stream; }
int flexflow_model_uses_graph(flexflow_model_t m) { return M(m)->config.enable_graph ? 1 : 0; }
void flexflow_model_set_trace_mode(flexflow_model_t m, int mode) { M(m)->config.trace_mode = mode; }
int flexflow_model_trace_replays(flexflow_model_t m, int trace_id) { return M(m)->trace_replays(trace_id) ? 1 : 0; }
int64_t flexflow_model_get_counter(flexflow_model_t m, const char* name) {
  const std::string n(name ? name : "");
  if (n == "mlp_chain_fwd_calls") return M(m)->n_chain_fwd_calls;
  if (n == "mlp_chain_bwd_calls") return M(m)->n_chain_bwd_calls;
  if (n == "allreduce_bucket_calls") return M(m)->n_bucket_allreduces;
  if (n == "allreduce_buckets") return (int64_t)M(m)->grad_buckets.size();
  if (n.rfind("allreduce_bucket_floats_", 0) == 0) {      // floats of bucket k, in issue order
    const size_t k = (size_t)atoi(n.c_str() + 24);
    return k < M(m)->grad_buckets.size() ? (int64_t)M(m)->grad_buckets[k].count : -1;
  }
  if (n == "allreduce_bucket_channel_own") return M(m)->config.comm.bucket_channel_own;
  if (n == "direct_allreduces") return M(m)->n_direct_allreduces;
  if (n == "tensor_op_exact_backward_layers") {      // Linear layers whose backward runs in exact mode under --allow-tensor-op-math-conversion (allocate() step 7)
    int64_t k = 0;
    for (Op* op : M(m)->layers) if (op->op_type == OP_LINEAR && static_cast<Linear*>(op)->bwd_exact) k++;
    return k;
  }
  return -1;
}

int flexflow_tensor_get_num_dims(flexflow_tensor_t t) { return T(t)->numDim; }
void flexflow_tensor_get_dims(flexflow_tensor_t t, int* dims) { for (int i = 0; i < T(t)->numDim; i++) dims[i] = T(t)->adim[T(t)->numDim - 1 - i]; }
int64_t flexflow_tensor_get_local_rows(flexflow_tensor_t t) { return T(t)->impl ? T(t)->impl->rows_local : 0; }
bool flexflow_tensor_is_local(flexflow_tensor_t t) { return T(t)->impl && T(t)->impl->ptr != nullptr; }
void* flexflow_tensor_get_device_ptr(flexflow_tensor_t t) { return T(t)->impl ? T(t)->impl->ptr : nullptr; }
int64_t flexflow_tensor_get_ld(flexflow_tensor_t t) { return T(t)->impl ? T(t)->impl->ld : 0; }
void flexflow_tensor_set_float(flexflow_tensor_t t, flexflow_model_t m, const int* dims, int nd, const float* d) { T(t)->set_tensor<float>(M(m), dims_vec(dims, nd), d); }
void flexflow_tensor_set_int64(flexflow_tensor_t t, flexflow_model_t m, const int* dims, int nd, const int64_t* d) { T(t)->set_tensor<int64_t>(M(m), dims_vec(dims, nd), d); }
void flexflow_tensor_get_float(flexflow_tensor_t t, flexflow_model_t m, float* d) { T(t)->get_tensor<float>(M(m), d); }
void flexflow_tensor_get_int64(flexflow_tensor_t t, flexflow_model_t m, int64_t* d) { T(t)->get_tensor<int64_t>(M(m), d); }
void flexflow_tensor_get_grad_float(flexflow_tensor_t t, flexflow_model_t m, float* d) { T(t)->get_grad<float>(M(m), d); }

flexflow_dlrm_t flexflow_dlrm_create(int argc, char** argv, const ffcomm* comm) { flexflow_dlrm_t h; h.impl = new DLRMApp(argc, argv, comm); return h; }
void flexflow_dlrm_destroy(flexflow_dlrm_t h) { delete A(h); }
flexflow_model_t flexflow_dlrm_get_model(flexflow_dlrm_t h) { flexflow_model_t m; m.impl = A(h)->ff; return m; }
int flexflow_dlrm_get_num_samples(flexflow_dlrm_t h) { return A(h)->loader->num_samples; }
int flexflow_dlrm_get_num_tables(flexflow_dlrm_t h) { return (int)A(h)->sparse_inputs.size(); }
flexflow_tensor_t flexflow_dlrm_get_sparse_input(flexflow_dlrm_t h, int t) { return wrap(A(h)->sparse_inputs.at(t)); }
flexflow_tensor_t flexflow_dlrm_get_dense_input(flexflow_dlrm_t h) { return wrap(A(h)->dense_input); }
void flexflow_dlrm_warmup(flexflow_dlrm_t h) { A(h)->warmup(); }
void flexflow_dlrm_train_steps(flexflow_dlrm_t h, int n, bool trace) { A(h)->train_steps(n, trace); }
double flexflow_dlrm_run_epochs(flexflow_dlrm_t h) { return A(h)->run_epochs(); }

// `iters` real eager steps with events around the side-stream gather / table update, the three collectives and the compute stream's
// wait for the embedding branch; out[k] = average milliseconds of pair k (FFModel::probe_ev: gather, update, forward all-to-all,
// backward all-to-all, all-reduce, join wait), 0 where a pair was never recorded.  COLLECTIVE at world_size > 1: the steps issue the
// exchange, so every rank must make this call (round-3 advisor: the probes of time_kernel(10 / 11) ran on rank 0 alone and hung).
void flexflow_dlrm_probe_step(flexflow_dlrm_t h, int iters, float* out, int nout) {
  DLRMApp* app = A(h);
  FFModel* ff = app->ff;
  for (int k = 0; k < nout; k++) out[k] = 0.0f;
  if (!app->warmed_up) app->warmup();
  ff->sync();
  ff->probe_events_on = true;
  const int npairs = FFModel::kProbeEvents / 2;
  std::vector<double> sum(npairs, 0.0);
  std::vector<int> cnt(npairs, 0);
  for (int i = 0; i < iters + 1; i++) {
    app->train_steps(1, false);
    ff->sync();
    if (i == 0) continue;                      // first step: warm
    for (int k = 0; k < npairs; k++) {
      if (!ff->probe_ev[2 * k] || !ff->probe_ev[2 * k + 1]) continue;
      float ms = 0.f;
      if (ff->api->ffh_event_elapsed_ms(ff->ctx, ff->probe_ev[2 * k], ff->probe_ev[2 * k + 1], &ms) == 0) { sum[k] += ms; cnt[k]++; }
    }
  }
  ff->probe_events_on = false;
  for (int k = 0; k < npairs && k < nout; k++) out[k] = cnt[k] ? (float)(sum[k] / cnt[k]) : 0.0f;
}

float flexflow_dlrm_time_kernel(flexflow_dlrm_t h, int which, int iters) {
  DLRMApp* app = A(h);
  FFModel* ff = app->ff;
  if (!app->warmed_up) app->warmup();
  ff->sync();
  if (which == 10 || which == 11) {
    // in-step probes: `iters` real eager steps with events around the side-stream gather (10) / table update (11); the average
    // of the event intervals = what those kernels take while they share the chip with the MLP kernels of the same step
    if (ff->embeddings.empty() || !ff->config.overlap_embedding) return 0.0f;
    ff->probe_events_on = true;
    double sum = 0.0; int n = 0;
    for (int i = 0; i < iters + 1; i++) {
      app->train_steps(1, false);
      ff->sync();
      const int a = which == 10 ? 0 : 2;
      if (i == 0 || !ff->probe_ev[a] || !ff->probe_ev[a + 1]) continue;     // first step: warm
      float ms = 0.f;
      if (ff->api->ffh_event_elapsed_ms(ff->ctx, ff->probe_ev[a], ff->probe_ev[a + 1], &ms) == 0) { sum += ms; n++; }
    }
    ff->probe_events_on = false;
    return n ? (float)(sum / n) : 0.0f;
  }
  // gather / update probes (8, 9): back-to-back launches rotate over kProbeSets id sets drawn with their own seeds, so that a
  // launch never finds the rows of the one before in the Infinity Cache (26 x 32768 rows x 512 B = 436 MB per set > 256 MiB)
  constexpr int kProbeSets = 4;
  std::vector<std::vector<const int64_t*>> probe_ids;
  std::vector<void*> probe_bufs;
  if (which == 8 || which == 9) {
    const int L = ff->embeddings.empty() ? 1 : ff->embeddings[0]->inputs[0].adim[0];
    for (int set = 0; set < kProbeSets; set++) {
      std::vector<const int64_t*> ptrs;
      for (const FFModel::EmbShard& sh : ff->shards) {
        if (sh.owner != ff->rank) continue;
        const int64_t n = (int64_t)ff->config.batchSize * L;
        void* buf = nullptr;
        ff->check(ff->api->ffh_malloc(ff->ctx, &buf, (size_t)n * sizeof(int64_t)), "probe ids");
        ff->check(ff->api->ffh_gen_indices(ff->ctx, (int64_t*)buf, n, 0x9e3779b97f4a7c15ull * (uint64_t)(set + 1) + (uint64_t)ptrs.size(), 0, sh.e->num_entries, ff->stream), "probe ids");
        probe_bufs.push_back(buf);
        ptrs.push_back((const int64_t*)buf);
      }
      probe_ids.push_back(ptrs);
    }
  }
  int probe_turn = 0;
  ffh_event e0, e1;
  ff->check(ff->api->ffh_event_create(ff->ctx, &e0), "event");
  ff->check(ff->api->ffh_event_create(ff->ctx, &e1), "event");
  auto body = [&](int n) {
    for (int i = 0; i < n; i++) {
      if (which == 6 || which == 7) {   // the Linear layer with the most multiply-adds: forward (6) / backward (7) alone
        Linear* big = nullptr;
        for (Op* op : ff->layers)
          if (Linear* li = dynamic_cast<Linear*>(op))
            if (!big || (double)li->in_channels * li->out_channels > (double)big->in_channels * big->out_channels) big = li;
        if (!big) continue;
        if (which == 6) big->forward(*ff); else big->backward(*ff);
        continue;
      }
      if (which == 8) ff->embedding_kernels_only(true, ff->stream, &probe_ids[probe_turn++ % kProbeSets]);          // gather kernels of this rank's shards, no exchange
      else if (which == 9) ff->embedding_kernels_only(false, ff->stream, &probe_ids[probe_turn++ % kProbeSets]);    // fused update kernels alone
      else if (which == 0) ff->embedding_group_forward(ff->stream);
      else if (which == 1) ff->embedding_group_update(ff->stream);
      else if (which == 3) {   // launch floor: a trivial dependent kernel (MSE gradient of the batch)
        const Tensor& fin = ff->layers.back()->outputs[0];
        ff->check(ff->api->ffh_mse_bwd(ff->ctx, fin.impl->grad, (const float*)fin.impl->ptr, (const float*)ff->label_tensor.impl->ptr,
                                       fin.impl->rows_local * fin.adim[0], 1.0f, ff->stream), "mse_bwd");
      } else if (which == 4) app->train_steps(1, false);   // eager step (no graph)
      else app->train_steps(1, true);
    }
  };
  body(2);   // warm
  ff->sync();
  if (which == 5) {
    // host side only: wall time to ENQUEUE one eager step (the GPU drains behind; the queue never blocks at this depth)
    which = 4;
    const auto t0 = std::chrono::steady_clock::now();
    body(iters);
    const auto t1 = std::chrono::steady_clock::now();
    ff->sync();
    ff->api->ffh_event_destroy(ff->ctx, e0);
    ff->api->ffh_event_destroy(ff->ctx, e1);
    return (float)(std::chrono::duration<double, std::milli>(t1 - t0).count() / iters);
  }
  ff->check(ff->api->ffh_event_record(ff->ctx, e0, ff->stream), "event");
  body(iters);
  if (which == 7 && ff->dw_forked) {   // the weight-gradient GEMM ran on its own stream: it belongs to the interval
    ff->check(ff->api->ffh_event_record(ff->ctx, ff->ev_dw_done, ff->dw_stream), "join dw");
    ff->check(ff->api->ffh_stream_wait_event(ff->ctx, ff->stream, ff->ev_dw_done), "join dw");
    ff->dw_forked = false;
    ff->dw1_used = false;
  }
  ff->check(ff->api->ffh_event_record(ff->ctx, e1, ff->stream), "event");
  ff->check(ff->api->ffh_event_sync(ff->ctx, e1), "event");
  ff->sync();
  float ms = 0.f;
  ff->check(ff->api->ffh_event_elapsed_ms(ff->ctx, e0, e1, &ms), "event");
  ff->api->ffh_event_destroy(ff->ctx, e0);
  ff->api->ffh_event_destroy(ff->ctx, e1);
  for (void* b : probe_bufs) ff->api->ffh_free(ff->ctx, b);
  return ms / (float)iters;
}

}  // extern "C"
