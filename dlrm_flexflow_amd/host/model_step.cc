// model_step.cc -- the training step
// (one of the translation units of the host shim: model_internal.h lists them)
#include "model_internal.h"

// =============================================================================================
// the training step [ref: src/runtime/model.cc:1410-1477, examples/cpp/DLRM/dlrm.cc:166-182]
// =============================================================================================
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
