/* rccl_comm.h -- the ffcomm callbacks implemented directly on RCCL (dlopen'ed), for GPU runs with world_size > 1.
 * The launcher still bootstraps: rank 0 makes a unique id (flexflow_rccl_get_unique_id), the launcher broadcasts the
 * 128 bytes over whatever it has (torch.distributed in bench.py), every rank calls flexflow_rccl_comm_create.
 * Replaces one ncclAllReduce per tensor [ref: src/runtime/optimizer_kernel.cu:170-171] with one per step, and the
 * Legion zero-copy movement of embedding outputs [ref: src/ops/embedding.cu:295-299] with an all-to-all made of
 * grouped ncclSend / ncclRecv pairs, enqueued on the model's own HIP streams (no host round trip per collective).
 * A row-wise sharded table (--row-shard-rows) adds ncclReduceScatter forward and ncclAllGather backward. */
#ifndef RCCL_COMM_H_
#define RCCL_COMM_H_
#include "ffcomm.h"
#ifdef __cplusplus
extern "C" {
#endif
/* lib_path: the RCCL to use (NULL: the one already loaded in the process, else librccl.so from the loader path).
 * Both return 0 on success; on failure nothing is left allocated and flexflow_rccl_last_error() says why. */
int  flexflow_rccl_available(const char* lib_path);      /* 0 = the library loads and has the symbols */
int  flexflow_rccl_get_unique_id(unsigned char id[128], const char* lib_path);
int  flexflow_rccl_comm_create(const unsigned char id[128], int rank, int world_size, const char* lib_path, ffcomm* out);
void flexflow_rccl_comm_destroy(ffcomm* comm);
const char* flexflow_rccl_last_error(void);
/* number of all-to-all / all-reduce calls served so far (tests) */
void flexflow_rccl_comm_calls(const ffcomm* comm, int64_t* alltoall, int64_t* allreduce);
void flexflow_rccl_comm_calls2(const ffcomm* comm, int64_t* reduce_scatter, int64_t* allgather);
/* gradient buckets (ffcomm.allreduce_bucket_sum_f32): calls served; *own_channel = 1 when they run on a second communicator */
int64_t flexflow_rccl_comm_bucket_calls(const ffcomm* comm, int* own_channel);
/* COLLECTIVE: a second communicator (ncclCommSplit) for the gradient buckets; 0 = it exists, 1 = not available (they share the first) */
int  flexflow_rccl_comm_enable_bucket_channel(ffcomm* comm);
/* LOCAL: 0 = this process's RCCL has ncclCommSplit (agree over the ranks before the collective call above); undo after a partial failure */
int  flexflow_rccl_has_comm_split(const char* lib_path);
void flexflow_rccl_comm_disable_bucket_channel(ffcomm* comm);
#ifdef __cplusplus
}
#endif
#endif
