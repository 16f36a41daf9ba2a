// model_allocate.cc -- FFModel::compile / allocate
// (one of the translation units of the host shim: model_internal.h lists them)
#include "model_internal.h"

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
  if (config.mlp_chain && !config.profiling && !config.async_launch && !config.allow_tensor_op_math_conversion) {      // (split mode: see the end of step 7)
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
  for (Op* op : layers) if (op->op_type == OP_LINEAR) { Linear* li = static_cast<Linear*>(op); li->dx_image = false; li->bwd_exact = false; li->dx_twin = nullptr; li->dx_twin_registered = false; }
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
      if (twin_linear(op)) {
        act_ok = true;
        if (x3_images) {      // 6 bytes per element in the epilogue: only where a layer the mode takes reads the image (its forward's x, its weight gradient's x)
          act_ok = false;
          for (Op* q : layers)
            for (int i = 0; i < q->numInputs; i++)
              if (q->inputs[i].impl == im && twin_linear(q)) act_ok = true;
        }
      }
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
      if (ncons == 1 && twin_read && only && (twin_linear(only) || skinny_twin(only)) && only->dx_overwrite && !only->discard_input_grad && im->grad && !exchange) {
        reg(im->grad, im->bytes, twin_at(grad_twin, (size_t)((const char*)im->grad - act_grad_slab)));
        if (!x3_images) {      // (bwd_exact below)
          Linear* o = const_cast<Linear*>(only);
          o->dx_twin_registered = true;
          if (im->grad_ld == op->outputs[0].adim[0]) o->dx_twin = twin_at(grad_twin, (size_t)((const char*)im->grad - act_grad_slab));
        }
      }
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
  // tensor-op mode: a SMALL layer whose dy arrives with a live activation derivative (the bottom MLP's last layer under the Concat: relu' cannot be
  // folded into the producer of dy) runs its backward in EXACT mode.  On the bf16 pipe that backward is a pass over dy (relu', bias gradient) and
  // two GEMMs on the converting 128 x 128 kernel, because dy, changed in place, has no valid twin: 256 -> 128 at 32768 samples 114 us alone against
  // 98 on the fp32 kernels (which fuse the mask), and 420 us of kernel time in the step's tail beside the table update.  The mode is per-ctx
  // state: backward() switches it around that one call, for whichever kernel library is loaded (the oracle follows the same host code).
  if (config.allow_tensor_op_math_conversion && !x3_images && config.bf16_exact_small_backward && !use_workers())
    for (Op* op : layers) {
      if (op->op_type != OP_LINEAR) continue;
      Linear* li = static_cast<Linear*>(op);
      const double flop = 2.0 * (double)local_rows(li->outputs[0], this) * (double)li->in_padded * (double)li->out_channels;
      if (li->in_channels < FFH_BF16_MIN_DIM || li->out_channels < FFH_BF16_MIN_DIM || flop >= 4.0e9) continue;
      if (li->dy_premasked || (li->activation != AC_MODE_RELU && li->activation != AC_MODE_SIGMOID)) continue;
      if (li->pair_upper || li->pair_lower || (li->dx_twin_registered && !li->dx_twin)) continue;
      li->bwd_exact = true;
    }
  // split mode: a chain of narrow layers (step 4e) runs as the chain launches only when none of its members keeps an image by a conversion of
  // its own behind its per-layer call (out_twin: forward(); dx_image: backward_part()) -- the chain launches would skip that pass.  (The library
  // refuses a chain with a member the mode's kernels would take: mlp_chain.hip, chain_math_mode_ok.)
  if (x3_images)
    for (Op* op : layers) {
      if (op->op_type != OP_LINEAR) continue;
      Linear* li = static_cast<Linear*>(op);
      auto converts = [](const std::vector<Linear*>& ch) { for (const Linear* m : ch) if (m->out_twin || m->dx_image) return true; return false; };
      if (converts(li->chain_fwd)) li->chain_fwd.clear();
      if (converts(li->chain_bwd)) li->chain_bwd.clear();
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
