/* ffcomm.h -- the collectives the data-parallel / table-wise-sharded DLRM step needs, as a POD
 * table of callbacks supplied by the launcher.
 *
 * The reference has no explicit exchange: embedding outputs are mapped to zero-copy host memory
 * and moved by Legion DMA [ref: src/ops/embedding.cu:295-299,376-381], and weight gradients are
 * summed by ncclAllReduce, one call per tensor [ref: src/runtime/optimizer_kernel.cu:170-171].
 * Here each process owns one GPU (one rank) and the launcher (bench.py / run_dlrm.py: torch.distributed
 * with the "nccl" backend = RCCL over xGMI; "gloo" in the CPU tests) provides:
 *   alltoall   uneven all-to-all of fp32 blocks  (embedding rows forward, their gradients backward)
 *   allreduce  in-place fp32 sum                 (the MLP gradients: one bucket per wide layer, issued as the backward produces them)
 *   reduce_scatter / allgather   (row-wise sharded giant table only, --row-shard-rows: every rank's partial bag sums
 *              for the global batch are summed and each rank keeps its own samples; the gradients of those
 *              samples are gathered back to every rank).  May be NULL when no table is row-sharded.
 * All calls are asynchronous on `stream` (a hipStream_t): they must be ordered after work already
 * enqueued on it and work enqueued later must see their result.  world_size == 1 needs no callbacks.
 */
#ifndef FFCOMM_H_
#define FFCOMM_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffcomm {
  int   rank;
  int   world_size;
  void* user;
  /* counts are in floats, one entry per peer; blocks are contiguous in rank order */
  int (*alltoall_f32)(void* user, const float* send, const int64_t* send_counts,
                      float* recv, const int64_t* recv_counts, void* stream);
  int (*allreduce_sum_f32)(void* user, float* buf, int64_t count, void* stream);
  int (*barrier)(void* user);
  /* 1: alltoall / allreduce only enqueue work on `stream` and return at once (RCCL called directly).  The model then
   * issues them in stream order; with 0 (a callback that takes tens of microseconds of host time, e.g. through an
   * interpreter) it first enqueues the kernels that do not depend on the collective. */
  int nonblocking;
  /* recv[i] = sum over ranks of their send[rank * recv_count + i]: send is [world_size * recv_count], recv is [recv_count] */
  int (*reduce_scatter_sum_f32)(void* user, const float* send, float* recv, int64_t recv_count, void* stream);
  /* recv[r * send_count + i] = rank r's send[i]: send is [send_count], recv is [world_size * send_count] */
  int (*allgather_f32)(void* user, const float* send, float* recv, int64_t send_count, void* stream);
  /* The same sum as allreduce_sum_f32 for ONE BUCKET of the MLP gradients, issued from inside backward() on the model's
   * communication stream while the rest of the backward runs (the reference issues one ncclAllReduce per parameter from that
   * parameter's own update task [ref: src/runtime/optimizer.cc:93-189, src/runtime/optimizer_kernel.cu:114-179]).  A transport that
   * can run it concurrently with the all-to-alls serves it on a channel of its own (RcclComm: a second communicator from
   * ncclCommSplit); NULL: the buckets go through allreduce_sum_f32. */
  int (*allreduce_bucket_sum_f32)(void* user, float* buf, int64_t count, void* stream);
  /* 1: allreduce_bucket_sum_f32 runs on a channel of its own -- the transport does NOT order a bucket against the all-to-alls, so the model issues
   * every bucket as soon as its layers have issued their backward.  0 (a shared channel: one RCCL communicator runs its collectives in issue
   * order whatever streams they are on): the model holds the buckets until the step's backward all-to-all has been enqueued, so that the
   * exchange of the embedding gradients -- and the table update and the next gather behind it -- never waits for a weight-gradient GEMM and
   * its all-reduce (FFModel::issue_grad_buckets). */
  int bucket_channel_own;
  /* The two collectives of the DIRECT all-reduce of a bucket (--direct-allreduce: all-to-all of 1 / world_size slices, local sum in rank
   * order, all-gather of the sums -- every link of the fully connected node carries 1 / world_size of the bucket, where a ring is bound by one
   * link), on the buckets' channel.  Same contracts as alltoall_f32 / allgather_f32.  NULL: the model uses those two. */
  int (*alltoall_bucket_f32)(void* user, const float* send, const int64_t* send_counts, float* recv, const int64_t* recv_counts, void* stream);
  int (*allgather_bucket_f32)(void* user, const float* send, float* recv, int64_t send_count, void* stream);
} ffcomm;

#ifdef __cplusplus
}
#endif
#endif
