// model_internal.h -- what the translation units of the host shim share beside ffmodel.h: the abort-with-message helper the reference's
// asserts become [ref: include/cuda_helper.h:6-47], small utilities, and the batched embedding launches (model_exchange.cc) the step uses.
//   model.cc           tensors, initializers, the operator classes, optimizers, FFModel's construction API
//   model_flags.cc     FFConfig: defaults and the command line [ref: src/runtime/model.cc:2212-2403]
//   model_allocate.cc  compile() / allocate(): storage, aliasing, shards, buckets, twins and images
//   model_exchange.cc  the embedding group (gather, exchange, fused update) and the MLP gradients' all-reduce (buckets, ring / direct)
//   model_step.cc      forward / zero_gradients / backward / update, traces, sync [ref: src/runtime/model.cc:1410-1477]
#pragma once
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

[[noreturn]] inline void die(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "FATAL: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  abort();   // the reference asserts/exits on every such condition [ref: include/cuda_helper.h:6-47]
}

inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline size_t dtype_size(DataType t) {
  switch (t) {
    case DT_FLOAT: return 4;
    case DT_DOUBLE: return 8;
    case DT_INT32: return 4;
    case DT_INT64: return 8;
    case DT_BOOLEAN: return 1;
    default: return 0;
  }
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// rows of a tensor this rank holds (the batch dimension is split over the ranks)
inline int64_t local_rows(const Tensor& t, const FFModel*) { return t.impl ? t.impl->rows_local : 0; }

// Parameters that live in the dense slab (one all-reduce bucket, one optimizer launch): Linear weights / biases and the
// tables of data-parallel (replicated) embeddings.
inline bool in_dense_slab(const Parameter& p) {
  if (p.owner_op->op_type == OP_LINEAR) return true;
  return p.owner_op->op_type == OP_EMBEDDING && static_cast<const Embedding*>(p.owner_op)->replicated;
}

// one batched kernel per distinct shard width of this rank's tables (model_exchange.cc): the gather, the fused backward + optimizer, or its
// two phases
enum ShardLaunch { kGather, kFusedUpdate, kSortOnly, kApplyOnly };
void launch_shard_groups(const FFModel* ff, ShardLaunch what, ffh_stream s, ffh_ctx* cx, const std::vector<const int64_t*>* idx_override = nullptr);
