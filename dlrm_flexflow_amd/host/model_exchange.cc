// model_exchange.cc -- the embedding group (gather, exchange, fused update) and the MLP gradients' all-reduce
// (one of the translation units of the host shim: model_internal.h lists them)
#include "model_internal.h"

// =============================================================================================
// embedding group: batched gather (+ exchange) and batched fused update
// =============================================================================================
// launches one batched kernel per distinct shard width (all table-wise tables share one; column blocks of
// giant tables another); FWD: gather, else fused backward + SGD
void launch_shard_groups(const FFModel* ff, ShardLaunch what, ffh_stream s, ffh_ctx* cx, const std::vector<const int64_t*>* idx_override) {
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
  // (round 6) in the two bf16-pipe math modes the GEMMs the update used to hide under are 1.5-4x shorter and the sort sits on the step's
  // critical chain [sort -> apply -> gather] at every batch: early there (32768 samples: split mode 5.21-5.27 vs 5.28-5.31 ms, tensor-op 1.949-1.960 vs 1.969-1.972)
  const bool exact_gemms = !config.allow_tensor_op_math_conversion && !config.fp32_split_bf16x3;
  const int mode = config.early_sort > 0 ? config.early_sort : ((!exchange && local_batch >= 8192 && exact_gemms) ? early_sort_big_batch_mode : 1);
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
  if (!config.direct_allreduce || !gather || !a2a || count <= 0) return ring();      // (one forced rank: the same three calls on one slice -- the functional run)
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
