// rccl_comm.cc -- see rccl_comm.h.  Only the stable core of the NCCL API is used; RCCL is found at run time.
#include "rccl_comm.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>

namespace {

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };

struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t*);
  int (*CommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int);
  int (*CommDestroy)(ncclComm_p);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  int (*Send)(const void*, size_t, int, int, ncclComm_p, void*);
  int (*Recv)(void*, size_t, int, int, ncclComm_p, void*);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, void*);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_p, void*);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_p, void*);
  const char* (*GetErrorString)(int);
  int (*CommSplit)(ncclComm_p, int, int, ncclComm_p*, void*);     // optional (NCCL >= 2.18 API)
};

std::string g_err;
Api g_api;

bool load(const char* path) {
  if (g_api.lib) return true;
  void* h = nullptr;
  if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // the copy the process already has (torch's)
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { g_err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
#define SYM(field, name)                                               \
  *(void**)(&g_api.field) = dlsym(h, name);                             \
  if (!g_api.field) { g_err = std::string("librccl lacks ") + name; return false; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(AllReduce, "ncclAllReduce") SYM(ReduceScatter, "ncclReduceScatter") SYM(AllGather, "ncclAllGather")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  *(void**)(&g_api.CommSplit) = dlsym(h, "ncclCommSplit");
  g_api.lib = h;
  return true;
}

struct Comm {
  ncclComm_p comm;
  int rank, world;
  int64_t n_alltoall, n_allreduce, n_reduce_scatter, n_allgather;
  ncclComm_p comm2;        // the gradient buckets' own channel (ncclCommSplit of comm), or null: they share comm
  int64_t n_bucket;
};

int fail(int rc, const char* what) {
  g_err = std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
  fprintf(stderr, "rccl_comm: %s\n", g_err.c_str());
  return 1;
}

// uneven all-to-all: one grouped send/recv pair per peer, blocks contiguous in rank order (counts in floats)
int alltoall_on(Comm* c, ncclComm_p comm, const float* send, const int64_t* sc, float* recv, const int64_t* rc, void* stream) {
  int e = g_api.GroupStart();
  if (e != kNcclSuccess) return fail(e, "ncclGroupStart");
  int64_t so = 0, ro = 0;
  for (int p = 0; p < c->world; p++) {
    if (sc[p] > 0) { e = g_api.Send(send + so, (size_t)sc[p], kNcclFloat32, p, comm, stream); if (e != kNcclSuccess) { g_api.GroupEnd(); return fail(e, "ncclSend"); } }
    if (rc[p] > 0) { e = g_api.Recv(recv + ro, (size_t)rc[p], kNcclFloat32, p, comm, stream); if (e != kNcclSuccess) { g_api.GroupEnd(); return fail(e, "ncclRecv"); } }
    so += sc[p]; ro += rc[p];
  }
  e = g_api.GroupEnd();
  if (e != kNcclSuccess) return fail(e, "ncclGroupEnd");
  return 0;
}
int alltoall_f32(void* user, const float* send, const int64_t* sc, float* recv, const int64_t* rc, void* stream) {
  Comm* c = (Comm*)user;
  if (alltoall_on(c, c->comm, send, sc, recv, rc, stream) != 0) return 1;
  c->n_alltoall++;
  return 0;
}
// the direct all-reduce's two collectives on the buckets' channel (the second communicator where there is one)
int alltoall_bucket_f32(void* user, const float* send, const int64_t* sc, float* recv, const int64_t* rc, void* stream) {
  Comm* c = (Comm*)user;
  return alltoall_on(c, c->comm2 ? c->comm2 : c->comm, send, sc, recv, rc, stream);
}
int allgather_bucket_f32(void* user, const float* send, float* recv, int64_t send_count, void* stream) {
  Comm* c = (Comm*)user;
  if (send_count > 0) {
    const int e = g_api.AllGather(send, recv, (size_t)send_count, kNcclFloat32, c->comm2 ? c->comm2 : c->comm, stream);
    if (e != kNcclSuccess) return fail(e, "ncclAllGather (bucket)");
  }
  return 0;
}

int allreduce_sum_f32(void* user, float* buf, int64_t count, void* stream) {
  Comm* c = (Comm*)user;
  if (count > 0) {
    const int e = g_api.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, stream);
    if (e != kNcclSuccess) return fail(e, "ncclAllReduce");
  }
  c->n_allreduce++;
  return 0;
}

// one bucket of the MLP gradients, issued from inside the backward on the model's communication stream: on the second communicator
// where there is one, so that RCCL does not order it against the all-to-alls of the first
int allreduce_bucket_sum_f32(void* user, float* buf, int64_t count, void* stream) {
  Comm* c = (Comm*)user;
  if (count > 0) {
    const int e = g_api.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm2 ? c->comm2 : c->comm, stream);
    if (e != kNcclSuccess) return fail(e, "ncclAllReduce (bucket)");
  }
  c->n_bucket++;
  return 0;
}

// row-wise sharded table, forward: partial bag sums of the global batch -> this rank's samples, summed over the ranks
int reduce_scatter_sum_f32(void* user, const float* send, float* recv, int64_t recv_count, void* stream) {
  Comm* c = (Comm*)user;
  if (recv_count > 0) {
    const int e = g_api.ReduceScatter(send, recv, (size_t)recv_count, kNcclFloat32, kNcclSum, c->comm, stream);
    if (e != kNcclSuccess) return fail(e, "ncclReduceScatter");
  }
  c->n_reduce_scatter++;
  return 0;
}

// ... backward: every rank's gradient rows -> every rank
int allgather_f32(void* user, const float* send, float* recv, int64_t send_count, void* stream) {
  Comm* c = (Comm*)user;
  if (send_count > 0) {
    const int e = g_api.AllGather(send, recv, (size_t)send_count, kNcclFloat32, c->comm, stream);
    if (e != kNcclSuccess) return fail(e, "ncclAllGather");
  }
  c->n_allgather++;
  return 0;
}

}  // namespace

extern "C" {

const char* flexflow_rccl_last_error(void) { return g_err.c_str(); }

int flexflow_rccl_available(const char* lib_path) { return load(lib_path) ? 0 : 1; }

int flexflow_rccl_get_unique_id(unsigned char id[128], const char* lib_path) {
  if (!load(lib_path)) return 1;
  ncclUniqueId_t u;
  const int e = g_api.GetUniqueId(&u);
  if (e != kNcclSuccess) return fail(e, "ncclGetUniqueId");
  memcpy(id, u.internal, 128);
  return 0;
}

int flexflow_rccl_comm_create(const unsigned char id[128], int rank, int world_size, const char* lib_path, ffcomm* out) {
  if (!out || rank < 0 || world_size < 1 || rank >= world_size) { g_err = "bad arguments"; return 1; }
  if (!load(lib_path)) return 1;
  ncclUniqueId_t u;
  memcpy(u.internal, id, 128);
  Comm* c = new Comm{nullptr, rank, world_size, 0, 0, 0, 0, nullptr, 0};
  const int e = g_api.CommInitRank(&c->comm, world_size, u, rank);
  if (e != kNcclSuccess) { delete c; return fail(e, "ncclCommInitRank"); }

  memset(out, 0, sizeof *out);
  out->rank = rank;
  out->world_size = world_size;
  out->user = c;
  out->alltoall_f32 = alltoall_f32;
  out->allreduce_sum_f32 = allreduce_sum_f32;
  out->barrier = nullptr;          // the launcher supplies its own (it owns the bootstrap group)
  out->nonblocking = 1;
  out->reduce_scatter_sum_f32 = reduce_scatter_sum_f32;
  out->allgather_f32 = allgather_f32;
  out->allreduce_bucket_sum_f32 = allreduce_bucket_sum_f32;
  out->alltoall_bucket_f32 = alltoall_bucket_f32;
  out->allgather_bucket_f32 = allgather_bucket_f32;
  return 0;
}

void flexflow_rccl_comm_destroy(ffcomm* comm) {
  if (!comm || !comm->user) return;
  Comm* c = (Comm*)comm->user;
  if (c->comm2) g_api.CommDestroy(c->comm2);
  if (c->comm) g_api.CommDestroy(c->comm);
  delete c;
  comm->user = nullptr;
}

void flexflow_rccl_comm_calls(const ffcomm* comm, int64_t* a2a, int64_t* ar) {
  const Comm* c = comm ? (const Comm*)comm->user : nullptr;
  if (a2a) *a2a = c ? c->n_alltoall : 0;
  if (ar) *ar = c ? c->n_allreduce : 0;
}

// A second communicator over the same ranks for the gradient buckets, so that RCCL does not order them against the all-to-alls of the
// first (it runs one communicator's collectives in issue order whatever streams they are on).  COLLECTIVE: every rank makes the call.
// The launchers' default where every rank's RCCL has ncclCommSplit (see flexflow_rccl_has_comm_split): the order problem it removes is structural,
// not a tuning question; --allreduce-shared-channel turns it off for A/B runs.  Returns 0 when the channel
// exists afterwards, 1 when the library has no ncclCommSplit or it failed (the buckets then share the first communicator).
int flexflow_rccl_comm_enable_bucket_channel(ffcomm* comm) {
  Comm* c = comm ? (Comm*)comm->user : nullptr;
  if (!c) return 1;
  if (c->comm2) return 0;
  if (!g_api.CommSplit) { g_err = "librccl has no ncclCommSplit"; return 1; }
  const int e = g_api.CommSplit(c->comm, 0, c->rank, &c->comm2, nullptr);
  if (e != kNcclSuccess) { c->comm2 = nullptr; return fail(e, "ncclCommSplit"); }
  comm->bucket_channel_own = 1;
  return 0;
}
// 0 when this process's RCCL has ncclCommSplit.  LOCAL: the launchers agree on it over all ranks BEFORE any of them makes the collective
// call above (a rank without the symbol would return at once and leave the others inside ncclCommSplit).
int flexflow_rccl_has_comm_split(const char* lib_path) { return (load(lib_path) && g_api.CommSplit) ? 0 : 1; }
// Back to the shared channel (local; every rank calls it when the ranks did not ALL get their second communicator: a rank with comm2 and
// a rank without would issue the same bucket on different communicators and hang).
void flexflow_rccl_comm_disable_bucket_channel(ffcomm* comm) {
  Comm* c = comm ? (Comm*)comm->user : nullptr;
  if (!c) return;
  if (c->comm2) { g_api.CommDestroy(c->comm2); c->comm2 = nullptr; }
  comm->bucket_channel_own = 0;
}

int64_t flexflow_rccl_comm_bucket_calls(const ffcomm* comm, int* own_channel) {
  const Comm* c = comm ? (const Comm*)comm->user : nullptr;
  if (own_channel) *own_channel = (c && c->comm2) ? 1 : 0;
  return c ? c->n_bucket : 0;
}

void flexflow_rccl_comm_calls2(const ffcomm* comm, int64_t* rs, int64_t* ag) {
  const Comm* c = comm ? (const Comm*)comm->user : nullptr;
  if (rs) *rs = c ? c->n_reduce_scatter : 0;
  if (ag) *ag = c ? c->n_allgather : 0;
}

}  // extern "C"
