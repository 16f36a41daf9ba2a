/* ffh_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the arithmetic of the reference's DLRM hot path
 * (facebookresearch/DLRM-FlexFlow), exporting the same C-ABI as the HIP library
 * (include/ff_hip.h) with "device" pointers being host pointers.  It exists so
 * that tests can check the HIP kernels against it, so that
 * __graft_entry__.smoke() can check one invocation, and so that bench.py can
 * time a CPU baseline ("cpu_baseline.kind" = "port").  Nothing in the product
 * path (dlrm_flexflow_amd/, the C++ FFModel shim's default backend) loads it.
 *
 * Pinning: the embedding forward is checked against the reference's own AVX2
 * lookup, and the embedding backward (dense scatter-add, and through it the
 * fused backward + SGD) against the reference's own CPU embed_backward, both
 * compiled from /root/reference (oracle/Makefile -> oracle/_ref/); the Linear /
 * Concat / BatchMatmul / SGD / Adam / MSE functions against PyTorch-CPU + numpy,
 * the oracle the reference's op tests use (tests/ops/test_harness.py); vectors
 * are committed under tests/golden/ (tests/golden/make_golden.py).
 * The pairwise-dot interaction (ffh_tril_*, ffh_dot_interaction_*) has no reference
 * operator at all (the driver's "dot" is a TODO, examples/cpp/DLRM/dlrm.cc:53-54): it
 * is pinned against torch.bmm + tril_indices, the composition the reference's op
 * tests spell (tests/ops/test_harness.py:125-177) plus MLPerf's triangle.  The fused
 * entry points ffh_linear_bwd_mse / ffh_linear_pair_bwd are, here, literally the two
 * calls each stands for.
 * The metrics kernel's accumulation order, bags with more than one index in the
 * backward, and the multi-GPU exchange have no reference function, test or
 * golden vector: PARITY UNPINNED by the reference for those -- they are checked
 * only against the float64 mathematical result / the single-rank run
 * (tests/test_oracle_golden.py, tests/test_ffmodel_host.py).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Summation orders are fixed and documented so the
 * results do not depend on the OpenMP thread count: threads only ever split
 * independent output elements.
 */
#include "../include/ff_hip.h"
#include "../include/ffh_rng.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct ffh_ctx {
  void*  ws;
  size_t ws_bytes;
  char   err[256];
  const ffh_col_dest* scatter_map;   /* ffh_linear_bwd_set_dx_scatter: pending for the next ffh_linear_bwd_ex */
  float* colsum_dst; int colsum_ncols, colsum_used;   /* ffh_linear_bwd_set_dx_colsum: pending for the next ffh_linear_bwd / _ex */
  int    scatter_ncols, scatter_used;
  int    math_mode;                  /* ffh_ctx_set_math_mode */
  struct { const char* base; size_t bytes; char* img; } x3[64];   /* ffh_ctx_bf16x3_mirror_set */
  int    nx3;
};

static int fail(ffh_ctx* c, int code, const char* msg) {
  if (c) snprintf(c->err, sizeof c->err, "%s", msg);
  return code;
}

/* ------------------------------------------------------------------ */
/* library / context / memory: host stand-ins                          */
/* ------------------------------------------------------------------ */
int         ffh_abi_version(void) { return FFH_ABI_VERSION; }
const char* ffh_backend_name(void) { return "oracle-cpu"; }

int ffh_ctx_create(ffh_ctx** out, int device) {
  (void)device;
  if (!out) return FFH_ERR_BAD_ARG;
  ffh_ctx* c = (ffh_ctx*)calloc(1, sizeof *c);
  if (!c) return FFH_ERR_NOMEM;
  *out = c;
  return FFH_OK;
}
int ffh_ctx_default(ffh_ctx** out) {          /* one library-owned ctx (the "device" is the host) */
  static ffh_ctx* def;
  if (!out) return FFH_ERR_BAD_ARG;
  if (!def) { const int rc = ffh_ctx_create(&def, 0); if (rc != FFH_OK) return rc; }
  *out = def;
  return FFH_OK;
}
int ffh_ctx_destroy(ffh_ctx* c) { free(c); return FFH_OK; }
/* bf16 twins are a bandwidth optimisation of the HIP library (the twin of x is bf16_round(x), which this oracle computes at
 * the point of use): registrations are accepted and ignored; the explicit conversion is the rounding itself */
int ffh_ctx_bf16_mirror_set(ffh_ctx* c, const void* base, size_t bytes, void* twin) { (void)twin; return (c && base && bytes) ? FFH_OK : FFH_ERR_BAD_ARG; }
const char* ffh_linear_last_route(const ffh_ctx* c) { (void)c; return "oracle"; }
const char* ffh_embedding_last_route(const ffh_ctx* c) { (void)c; return "oracle"; }
const char* ffh_last_error_string(const ffh_ctx* c) { return c ? c->err : "null ctx"; }
int ffh_device_query(ffh_ctx* c, ffh_device_info* info) {
  if (!c || !info) return FFH_ERR_BAD_ARG;
  memset(info, 0, sizeof *info);
  snprintf(info->name, sizeof info->name, "host CPU (oracle)");
  snprintf(info->arch, sizeof info->arch, "cpu");
  info->wavefront_size = 1;
  return FFH_OK;
}
int ffh_ctx_set_workspace(ffh_ctx* c, void* ws, size_t bytes) {
  if (!c) return FFH_ERR_BAD_ARG;
  c->ws = ws; c->ws_bytes = bytes;
  return FFH_OK;
}
/* cublasSetMathMode(CUBLAS_TENSOR_OP_MATH) [ref: src/runtime/model.cu:81-83]: see include/ff_hip.h for what it selects */
int ffh_ctx_set_math_mode(ffh_ctx* c, int mode) {
  if (!c || (mode != FFH_MATH_DEFAULT && mode != FFH_MATH_TENSOR_OP_BF16 && mode != FFH_MATH_FP32_SPLIT_BF16X3 && mode != FFH_MATH_FP32_SPLIT_BF16X3_ALL)) return FFH_ERR_BAD_ARG;
  c->math_mode = mode;
  return FFH_OK;
}
/* the oracle's sums are sequential: it is deterministic in either mode */
int ffh_ctx_set_deterministic(ffh_ctx* c, int on) { (void)on; return c ? FFH_OK : FFH_ERR_BAD_ARG; }
/* the oracle needs no scratch */
int ffh_ctx_reserve_scratch(ffh_ctx* c, ffh_stream s) { (void)s; return c ? FFH_OK : FFH_ERR_BAD_ARG; }
int ffh_ctx_set_dw_cu_reserve(ffh_ctx* c, int ncus) { return (c && ncus >= 0) ? FFH_OK : FFH_ERR_BAD_ARG; }   /* a scheduling hint: nothing to do on the host */
/* float -> bfloat16 (round to nearest even, NaN kept) -> float: what v_cvt_pk_bf16_f32 leaves in the MFMA operand */
static inline float bf16_round(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;          /* NaN stays a (quiet) NaN */
  else u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&v, &u, 4);
  return v;
}
int ffh_convert_f32_to_bf16(ffh_ctx* c, void* dst, const float* src, int64_t n, ffh_stream s) {
  (void)s;
  if (!c || n < 0 || ((!dst || !src) && n)) return FFH_ERR_BAD_ARG;
  uint16_t* d = (uint16_t*)dst;
  for (int64_t i = 0; i < n; i++) { const float r = bf16_round(src[i]); uint32_t u; memcpy(&u, &r, 4); d[i] = (uint16_t)(u >> 16); }
  return FFH_OK;
}
/* Three-plane images of the split mode (ff_hip.h, "I32": 32 fp32 elements <-> 192 bytes [x1 | x2 | x3], x1 = bf16(x), x2 = bf16(x - x1),
 * x3 = bf16(x - x1 - x2)).  The oracle's GEMMs never read them (its split mode is the fp32 arithmetic the mode is held to); registrations are
 * kept so that ffh_convert_f32_to_bf16x3 can restate the image on the host -- the tests compare the HIP producers' images with it bit for bit. */
int ffh_ctx_bf16x3_mirror_set(ffh_ctx* c, const void* base, size_t bytes, void* img) {
  if (!c || !base || bytes == 0 || ((uintptr_t)base & 127) || ((uintptr_t)img & 127)) return FFH_ERR_BAD_ARG;
  int at = -1;
  for (int i = 0; i < c->nx3; i++) if (c->x3[i].base == (const char*)base) at = i;
  if (!img) { if (at >= 0) c->x3[at] = c->x3[--c->nx3]; return FFH_OK; }
  if (at < 0) { if (c->nx3 >= 64) return fail(c, FFH_ERR_UNSUPPORTED, "bf16x3_mirror_set: more than 64 regions"); at = c->nx3++; }
  c->x3[at].base = (const char*)base; c->x3[at].bytes = bytes; c->x3[at].img = (char*)img;
  return FFH_OK;
}
int ffh_convert_f32_to_bf16x3(ffh_ctx* c, const float* src, int64_t rows, int64_t cols, int64_t ld, ffh_stream s) {
  (void)s;
  if (!c || rows < 0 || cols < 0 || (rows > 1 && ld < cols)) return FFH_ERR_BAD_ARG;
  if (rows == 0 || cols == 0) return FFH_OK;
  if (!src) return FFH_ERR_BAD_ARG;
  const char* q = (const char*)src;
  const size_t span = (size_t)((rows - 1) * ld + cols) * 4;
  for (int i = 0; i < c->nx3; i++) {
    if (q < c->x3[i].base || q + span > c->x3[i].base + c->x3[i].bytes) continue;
    const int64_t e0 = (int64_t)(q - c->x3[i].base) / 4;
    for (int64_t r = 0; r < rows; r++)
      for (int64_t k = 0; k < cols; k++) {
        const float x = src[r * ld + k];
        const float x1 = bf16_round(x), r1 = x - x1, x2 = bf16_round(r1), r2 = r1 - x2, x3 = bf16_round(r2);
        const int64_t e = e0 + r * ld + k;
        uint16_t* d = (uint16_t*)(c->x3[i].img + (e >> 5) * 192 + (e & 31) * 2);
        uint32_t u;
        memcpy(&u, &x1, 4); d[0] = (uint16_t)(u >> 16);
        memcpy(&u, &x2, 4); d[32] = (uint16_t)(u >> 16);
        memcpy(&u, &x3, 4); d[64] = (uint16_t)(u >> 16);
      }
    return FFH_OK;
  }
  return fail(c, FFH_ERR_BAD_ARG, "convert_f32_to_bf16x3: not inside a region registered with ffh_ctx_bf16x3_mirror_set");
}
static inline int use_bf16(const ffh_ctx* c, int in, int out) {
  return c && c->math_mode == FFH_MATH_TENSOR_OP_BF16 && in >= FFH_BF16_MIN_DIM && out >= FFH_BF16_MIN_DIM;
}
int ffh_malloc(ffh_ctx* c, void** p, size_t bytes) {
  if (!c || !p) return FFH_ERR_BAD_ARG;
  *p = NULL;
  if (posix_memalign(p, 256, bytes ? bytes : 256)) return fail(c, FFH_ERR_NOMEM, "host alloc failed");
  return FFH_OK;
}
int ffh_free(ffh_ctx* c, void* p) { (void)c; free(p); return FFH_OK; }
int ffh_memcpy_h2d(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) { (void)c; (void)st; memcpy(d, s, n); return FFH_OK; }
int ffh_memcpy_d2h(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) { (void)c; (void)st; memcpy(d, s, n); return FFH_OK; }
int ffh_memcpy_d2d(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) { (void)c; (void)st; memmove(d, s, n); return FFH_OK; }
int ffh_stream_create(ffh_ctx* c, ffh_stream* s) { (void)c; if (s) *s = NULL; return FFH_OK; }
int ffh_stream_create_with_priority(ffh_ctx* c, ffh_stream* s, int priority) { (void)c; (void)priority; if (!s) return FFH_ERR_BAD_ARG; *s = NULL; return FFH_OK; }
int ffh_stream_destroy(ffh_ctx* c, ffh_stream s) { (void)c; (void)s; return FFH_OK; }
int ffh_stream_sync(ffh_ctx* c, ffh_stream s) { (void)c; (void)s; return FFH_OK; }
int ffh_device_sync(ffh_ctx* c) { (void)c; return FFH_OK; }
/* events carry a host timestamp so that ffh_event_elapsed_ms works */
int ffh_event_create(ffh_ctx* c, ffh_event* e) {
  (void)c; if (!e) return FFH_ERR_BAD_ARG;
  *e = calloc(1, sizeof(double));
  return *e ? FFH_OK : FFH_ERR_NOMEM;
}
int ffh_event_create_sync(ffh_ctx* c, ffh_event* e) {
  (void)c; if (!e) return FFH_ERR_BAD_ARG;
  *e = calloc(1, sizeof(double));
  return *e ? FFH_OK : FFH_ERR_NOMEM;
}
int ffh_event_destroy(ffh_ctx* c, ffh_event e) { (void)c; free(e); return FFH_OK; }
int ffh_event_record(ffh_ctx* c, ffh_event e, ffh_stream s) {
  (void)c; (void)s; if (!e) return FFH_ERR_BAD_ARG;
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  *(double*)e = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return FFH_OK;
}
int ffh_event_sync(ffh_ctx* c, ffh_event e) { (void)c; (void)e; return FFH_OK; }
int ffh_stream_wait_event(ffh_ctx* c, ffh_stream s, ffh_event e) { (void)c; (void)s; (void)e; return FFH_OK; }
int ffh_event_elapsed_ms(ffh_ctx* c, ffh_event a, ffh_event b, float* ms) {
  (void)c; if (!a || !b || !ms) return FFH_ERR_BAD_ARG;
  *ms = (float)(*(double*)b - *(double*)a);
  return FFH_OK;
}
/* no graphs on the host: capture is refused so callers run eagerly */
int ffh_graph_begin_capture(ffh_ctx* c, ffh_stream s) { (void)s; return fail(c, FFH_ERR_UNSUPPORTED, "oracle: no graph capture"); }
int ffh_graph_end_capture(ffh_ctx* c, ffh_stream s, ffh_graph* g) { (void)s; (void)g; return fail(c, FFH_ERR_UNSUPPORTED, "oracle: no graph capture"); }
int ffh_graph_launch(ffh_ctx* c, ffh_graph g, ffh_stream s) { (void)g; (void)s; return fail(c, FFH_ERR_UNSUPPORTED, "oracle: no graph capture"); }
int ffh_graph_destroy(ffh_ctx* c, ffh_graph g) { (void)c; (void)g; return FFH_OK; }

/* ------------------------------------------------------------------ */
/* initialisers / synthetic data                                       */
/* ------------------------------------------------------------------ */
/* assign_kernel [ref: src/runtime/cuda_helper.cu:52-60] */
int ffh_fill_f32(ffh_ctx* c, float* p, int64_t n, float v, ffh_stream s) {
  (void)c; (void)s;
  for (int64_t i = 0; i < n; i++) p[i] = v;
  return FFH_OK;
}
int ffh_zero(ffh_ctx* c, void* p, size_t bytes, ffh_stream s) { (void)c; (void)s; memset(p, 0, bytes); return FFH_OK; }
/* UniformInitializer::init_task [ref: src/runtime/initializer_kernel.cu:24-98]: curandGenerateUniform
 * then scale_kernel(min,max); here the uniform comes from include/ffh_rng.h */
int ffh_init_uniform(ffh_ctx* c, float* p, int64_t n, uint64_t seed, float lo, float hi, ffh_stream s) {
  (void)c; (void)s;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) p[i] = ffh_uniform(ffh_hash(seed, (uint64_t)i), lo, hi);
  return FFH_OK;
}
/* DataLoader::load_entire_dataset synthetic branch [ref: examples/cpp/DLRM/dlrm.cc:413-420] */
int ffh_gen_indices(ffh_ctx* c, int64_t* idx, int64_t n, uint64_t seed, int64_t first, int64_t R, ffh_stream s) {
  (void)s;
  if (R <= 0) return fail(c, FFH_ERR_BAD_ARG, "gen_indices: num_entries <= 0");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) idx[i] = ffh_index(ffh_hash(seed, (uint64_t)(first + i)), R);
  return FFH_OK;
}
int ffh_gen_uniform01(ffh_ctx* c, float* p, int64_t n, uint64_t seed, int64_t first, ffh_stream s) {
  (void)c; (void)s;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) p[i] = ffh_u24(ffh_hash(seed, (uint64_t)(first + i)));
  return FFH_OK;
}
int ffh_gen_bernoulli(ffh_ctx* c, float* p, int64_t n, uint64_t seed, int64_t first, ffh_stream s) {
  (void)c; (void)s;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) p[i] = ffh_bernoulli(ffh_hash(seed, (uint64_t)(first + i)));
  return FFH_OK;
}

/* ------------------------------------------------------------------ */
/* Embedding                                                          */
/* ------------------------------------------------------------------ */
/* embed_forward [ref: src/ops/embedding.cu:166-190]: output[i] = 0; for j ascending
 * output[i] += embed[idx*out_dim + off].  The CPU twin
 * [ref: src/ops/embedding.cc:262-314] does op[j] = fma(1.0f, ip[j], op[j]) from a zeroed
 * op, which is the same value.  AVG: the reference divides inside the j loop (a bug,
 * SURVEY 8a-1); the restated AVG is the true mean sum * (1.0f/in_dim), as the CPU
 * twin's normalize_by_lengths branch computes it [ref: src/ops/embedding.cc:300-311]. */
int ffh_embedding_fwd(ffh_ctx* c, const int64_t* idx, float* out, const float* w,
                      int L, int D, int64_t B, int64_t R, int64_t out_ld, int aggr, ffh_stream s) {
  (void)s;
  if (L <= 0 || D <= 0 || B < 0 || out_ld < D) return fail(c, FFH_ERR_BAD_ARG, "embedding_fwd: bad dims");
  if (aggr != FFH_AGGR_MODE_SUM && aggr != FFH_AGGR_MODE_AVG) return fail(c, FFH_ERR_BAD_ARG, "embedding_fwd: aggr");
  for (int64_t i = 0; i < B * L; i++)
    if (idx[i] < 0 || idx[i] >= R) return fail(c, FFH_ERR_BAD_ARG, "embedding_fwd: index out of range");
  const float inv = 1.0f / (float)L;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    float* o = out + b * out_ld;
    for (int d = 0; d < D; d++) o[d] = 0.0f;
    for (int j = 0; j < L; j++) {
      const float* row = w + idx[b * L + j] * (int64_t)D;
      for (int d = 0; d < D; d++) o[d] = o[d] + row[d];
    }
    if (aggr == FFH_AGGR_MODE_AVG)
      for (int d = 0; d < D; d++) o[d] = o[d] * inv;
  }
  return FFH_OK;
}

int ffh_embedding_fwd_multi(ffh_ctx* c, const ffh_emb_table* t, int nt, int L, int D, int64_t B, int aggr, ffh_stream s) {
  if (nt < 0 || nt > FFH_MAX_TABLES) return fail(c, FFH_ERR_BAD_ARG, "embedding_fwd_multi: ntables");
  for (int i = 0; i < nt; i++) {
    int rc = ffh_embedding_fwd(c, t[i].idx, t[i].io, t[i].weight, L, D, B, t[i].num_entries, t[i].ld, aggr, s);
    if (rc) return rc;
  }
  return FFH_OK;
}

/* embed_backward [ref: src/ops/embedding.cu:192-217]: atomicAdd(embed + idx*D + off, g) for
 * every (b, j).  Atomic arrival order is undefined in the reference; restated in the order
 * (b ascending, j ascending), which is what a single-threaded run of the kernel gives. */
int ffh_embedding_bwd_dense(ffh_ctx* c, const int64_t* idx, const float* g, float* wg,
                            int L, int D, int64_t B, int64_t R, int64_t gld, int aggr, ffh_stream s) {
  (void)s;
  if (L <= 0 || D <= 0 || B < 0 || gld < D) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_dense: bad dims");
  if (aggr != FFH_AGGR_MODE_SUM && aggr != FFH_AGGR_MODE_AVG) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_dense: aggr");
  for (int64_t i = 0; i < B * L; i++)
    if (idx[i] < 0 || idx[i] >= R) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_dense: index out of range");
  for (int64_t b = 0; b < B; b++)
    for (int j = 0; j < L; j++) {
      float* row = wg + idx[b * L + j] * (int64_t)D;
      for (int d = 0; d < D; d++) {
        float gr = g[b * gld + d];
        if (aggr == FFH_AGGR_MODE_AVG) gr = gr / (float)L;   /* [ref: embedding.cu:206-209] */
        row[d] += gr;
      }
    }
  return FFH_OK;
}

typedef struct { int64_t row; int64_t pos; } rp_t;
static int rp_cmp(const void* a, const void* b) {
  const rp_t* x = (const rp_t*)a; const rp_t* y = (const rp_t*)b;
  if (x->row != y->row) return x->row < y->row ? -1 : 1;
  if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
  return 0;
}

/* Net effect on an embedding table of one reference training step
 *   zero_grad   [ref: src/runtime/model.cc:466-490]  (dense gradient := 0)
 *   embed_backward [ref: src/ops/embedding.cu:192-217] (scatter-add)
 *   sgd_update  [ref: src/runtime/optimizer_kernel.cu:23-41] with momentum = 0, wd = 0
 *               (the driver's SGDOptimizer(&ff, 0.01f), [ref: examples/cpp/DLRM/dlrm.cc:130])
 * i.e. W[r] -= lr * sum of the gradients that hit r; rows not hit keep their bits
 * (W -= lr*0).  `W[i] -= lr*gt` is contracted to one FMA by nvcc's default -fmad=true, so
 * the update is restated as fmaf(-lr, sum, w).  The summation order is the canonical one
 * documented at ffh_embedding_bwd_sgd_fused in include/ff_hip.h. */
/* `opt` (may be NULL = plain SGD with lr): the rule applied to a touched row once its canonical gradient sum `tot` is complete
 * (ffh_sparse_opt, include/ff_hip.h) -- the element statements of ffh_sgd_update_ex / ffh_adam_update below, on the touched rows
 * only; s0 / s1 the per-row optimizer state. */
static int emb_bwd_opt_one(ffh_ctx* c, const int64_t* idx, const float* g, float* w, float* s0, float* s1,
                           int L, int D, int64_t B, int64_t R, int64_t gld, int aggr, float lr, const ffh_sparse_opt* opt) {
  if (L <= 0 || D <= 0 || B < 0 || gld < D) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_sgd_fused: bad dims");
  if (aggr != FFH_AGGR_MODE_SUM && aggr != FFH_AGGR_MODE_AVG) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_sgd_fused: aggr");
  const int64_t N = B * L;
  if (N == 0) return FFH_OK;
  rp_t* v = (rp_t*)malloc(sizeof(rp_t) * (size_t)N);
  float* part = (float*)malloc(sizeof(float) * (size_t)D);
  float* mid = (float*)malloc(sizeof(float) * (size_t)D);
  float* tot = (float*)malloc(sizeof(float) * (size_t)D);
  if (!v || !part || !mid || !tot) { free(v); free(part); free(mid); free(tot); return fail(c, FFH_ERR_NOMEM, "oom"); }
  for (int64_t p = 0; p < N; p++) {
    if (idx[p] < 0 || idx[p] >= R) { free(v); free(part); free(mid); free(tot); return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_sgd_fused: index out of range"); }
    v[p].row = idx[p]; v[p].pos = p;
  }
  qsort(v, (size_t)N, sizeof(rp_t), rp_cmp);
  const float invL = 1.0f / (float)L; (void)invL;
  int64_t i = 0;
  while (i < N) {
    const int64_t row = v[i].row;
    int64_t e = i;
    while (e < N && v[e].row == row) e++;
    /* two-level canonical order: 32-blocks inside 1024-blocks, left-to-right folds at every level */
    int first_big = 1;
    int64_t a1 = i;
    while (a1 < e) {
      int64_t z1 = (a1 / FFH_EMB_CHUNK1 + 1) * FFH_EMB_CHUNK1;
      if (z1 > e) z1 = e;
      int first_small = 1;
      int64_t a = a1;
      while (a < z1) {
        int64_t z = (a / FFH_EMB_CHUNK + 1) * FFH_EMB_CHUNK;
        if (z > z1) z = z1;
        for (int64_t q = a; q < z; q++) {
          const float* gr = g + (v[q].pos / L) * gld;
          for (int d = 0; d < D; d++) {
            float x = gr[d];
            if (aggr == FFH_AGGR_MODE_AVG) x = x / (float)L;
            part[d] = (q == a) ? x : part[d] + x;
          }
        }
        for (int d = 0; d < D; d++) mid[d] = first_small ? part[d] : mid[d] + part[d];
        first_small = 0;
        a = z;
      }
      for (int d = 0; d < D; d++) tot[d] = first_big ? mid[d] : tot[d] + mid[d];
      first_big = 0;
      a1 = z1;
    }
    float* wr = w + row * (int64_t)D;
    if (!opt || opt->kind == FFH_SPARSE_OPT_SGD) {
      for (int d = 0; d < D; d++) wr[d] = fmaf(-lr, tot[d], wr[d]);
    } else if (opt->kind == FFH_SPARSE_OPT_SGD_MOMENTUM) {      /* sgd_update [ref: src/runtime/optimizer_kernel.cu:23-41] on this row */
      float* vr = s0 ? s0 + row * (int64_t)D : NULL;
      for (int d = 0; d < D; d++) {
        float gt = fmaf(opt->weight_decay, wr[d], tot[d]);
        if (opt->momentum > 0.0f) {
          vr[d] = fmaf(vr[d], opt->momentum, gt);
          if (opt->nesterov) gt = fmaf(opt->momentum, vr[d], gt); else gt = vr[d];
        }
        wr[d] = fmaf(-opt->lr, gt, wr[d]);
      }
    } else {                                                     /* adam_update [ref: src/runtime/optimizer_kernel.cu:206-226] on this row */
      float* mr = s0 + row * (int64_t)D;
      float* vr = s1 + row * (int64_t)D;
      const float omb1 = 1.0f - opt->beta1, omb2 = 1.0f - opt->beta2;
      for (int d = 0; d < D; d++) {
        const float gt = fmaf(opt->weight_decay, wr[d], tot[d]);
        const float mt = fmaf(opt->beta1, mr[d], omb1 * gt);
        const float vt = fmaf(opt->beta2, vr[d], (omb2 * gt) * gt);
        mr[d] = mt;
        vr[d] = vt;
        wr[d] = wr[d] - (opt->lr * mt) / (sqrtf(vt) + opt->epsilon);
      }
    }
    i = e;
  }
  free(v); free(part); free(mid); free(tot);
  return FFH_OK;
}

int ffh_embedding_bwd_sgd_fused(ffh_ctx* c, const int64_t* idx, const float* g, float* w,
                                int L, int D, int64_t B, int64_t R, int64_t gld, int aggr, float lr, ffh_stream s) {
  (void)s;
  return emb_bwd_opt_one(c, idx, g, w, NULL, NULL, L, D, B, R, gld, aggr, lr, NULL);
}

/* ABI 8: the sparse (touched-rows) optimizers -- see ffh_sparse_opt in include/ff_hip.h for the stated lazy semantics */
int ffh_embedding_bwd_opt_fused_multi(ffh_ctx* c, const ffh_emb_table* t, const ffh_emb_state* st, int nt, int L, int D, int64_t B,
                                      int aggr, const ffh_sparse_opt* opt, ffh_stream s) {
  (void)s;
  if (!opt) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt_fused_multi: null ffh_sparse_opt");
  if (nt < 0 || nt > FFH_MAX_TABLES) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt_fused_multi: ntables");
  if (opt->kind != FFH_SPARSE_OPT_SGD && opt->kind != FFH_SPARSE_OPT_SGD_MOMENTUM && opt->kind != FFH_SPARSE_OPT_ADAM)
    return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: unknown ffh_sparse_opt.kind");
  if (opt->kind == FFH_SPARSE_OPT_SGD && (opt->weight_decay != 0.0f || opt->momentum != 0.0f))
    return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: FFH_SPARSE_OPT_SGD takes no weight decay / momentum");
  const int need0 = opt->kind == FFH_SPARSE_OPT_ADAM || (opt->kind == FFH_SPARSE_OPT_SGD_MOMENTUM && opt->momentum > 0.0f);
  for (int i = 0; i < nt; i++) {
    if (B > 0 && need0 && (!st || !st[i].s0)) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: optimizer state (s0) missing");
    if (B > 0 && opt->kind == FFH_SPARSE_OPT_ADAM && !st[i].s1) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: optimizer state (s1) missing");
    int rc = emb_bwd_opt_one(c, t[i].idx, t[i].io, t[i].weight, st ? st[i].s0 : NULL, st ? st[i].s1 : NULL, L, D, B, t[i].num_entries, t[i].ld, aggr, opt->lr, opt);
    if (rc) return rc;
  }
  return FFH_OK;
}
int ffh_embedding_bwd_opt_apply_multi(ffh_ctx* c, const ffh_emb_table* t, const ffh_emb_state* st, int nt, int L, int D, int64_t B,
                                      int aggr, const ffh_sparse_opt* opt, ffh_stream s) {
  return ffh_embedding_bwd_opt_fused_multi(c, t, st, nt, L, D, B, aggr, opt, s);
}

int ffh_embedding_bwd_sgd_fused_multi(ffh_ctx* c, const ffh_emb_table* t, int nt, int L, int D, int64_t B, int aggr, float lr, ffh_stream s) {
  if (nt < 0 || nt > FFH_MAX_TABLES) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_sgd_fused_multi: ntables");
  for (int i = 0; i < nt; i++) {
    int rc = ffh_embedding_bwd_sgd_fused(c, t[i].idx, t[i].io, t[i].weight, L, D, B, t[i].num_entries, t[i].ld, aggr, lr, s);
    if (rc) return rc;
  }
  return FFH_OK;
}

/* the two-call form (include/ff_hip.h, ABI 7): the index-only phase has nothing to precompute here -- the restatement sorts
 * inside the update -- so the sort call only validates and the apply call is the whole update */
int ffh_embedding_bwd_sort_multi(ffh_ctx* c, const ffh_emb_table* t, int nt, int L, int D, int64_t B, ffh_stream s) {
  (void)s;
  if (nt < 0 || nt > FFH_MAX_TABLES || (nt > 0 && !t) || L <= 0 || D <= 0 || B < 0) return fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_sort_multi: bad args");
  return FFH_OK;
}

int ffh_embedding_bwd_sgd_apply_multi(ffh_ctx* c, const ffh_emb_table* t, int nt, int L, int D, int64_t B, int aggr, float lr, ffh_stream s) {
  return ffh_embedding_bwd_sgd_fused_multi(c, t, nt, L, D, B, aggr, lr, s);
}
/* row-wise sharded table (this build's extension; include/ff_hip.h): ids of rows held elsewhere -> the zero row */
int ffh_embedding_localize_rows(ffh_ctx* c, const int64_t* idx, int64_t* local, int64_t n, int64_t row_begin, int64_t rows_local, ffh_stream s) {
  (void)s;
  if (n < 0 || row_begin < 0 || rows_local < 0 || (n > 0 && (!idx || !local))) return fail(c, FFH_ERR_BAD_ARG, "embedding_localize_rows: bad args");
  for (int64_t i = 0; i < n; i++) {
    const int64_t r = idx[i] - row_begin;
    local[i] = (r >= 0 && r < rows_local) ? r : rows_local;
  }
  return FFH_OK;
}

size_t ffh_embedding_bwd_workspace_bytes(int nt, int L, int D, int64_t B) { (void)nt; (void)L; (void)D; (void)B; return 0; }

/* ------------------------------------------------------------------ */
/* Linear                                                             */
/* ------------------------------------------------------------------ */
static float act_fwd(float v, int act) {
  if (act == FFH_AC_MODE_RELU) return v > 0.0f ? v : 0.0f;        /* CUDNN_ACTIVATION_RELU */
  if (act == FFH_AC_MODE_SIGMOID) return 1.0f / (1.0f + expf(-v)); /* CUDNN_ACTIVATION_SIGMOID */
  if (act == FFH_AC_MODE_GELU) {                                   /* gelu_forward_kernel [ref: src/runtime/cuda_helper.cu:81-90] */
    const float B = 0.7978845608028654f, C = 0.035677408136300125f;
    return v * (0.5f + 0.5f * tanhf(v * (C * v * v + B)));
  }
  return v;
}

/* the padding rule of the fast-path contract (include/ff_hip.h): the same answer as the product library */
int ffh_linear_fast_in_dim(int in, int out) {
  if (in <= 0 || out <= 0) return in;
  if (in % 64 == 0 || in < 256 || out % 128 != 0) return in;
  return (in + 63) / 64 * 64;
}

/* Linear::forward_kernel [ref: src/ops/linear.cu:436-453]:
 *   cublasSgemm(T,N, out,B,in): Y = W^T-view * X, beta 0      -> y[b][o] = sum_i x[b][i] w[o][i]
 *   cublasSgemm(T,N, out,B,1) with the ones vector, beta 1    -> y[b][o] += bias[o]
 *   cudnnActivationForward in place                           -> y = act(y)
 * cuBLAS's internal summation order is not published; restated as an i-ascending fp32
 * FMA chain from 0 (what v_mfma_f32_32x32x2_f32 also computes). */
int ffh_linear_fwd(ffh_ctx* c, const float* x, int64_t ldx, float* y, int64_t ldy,
                   const float* w, const float* bias, int in, int out, int64_t B, int act, ffh_stream s) {
  (void)s;
  if (in <= 0 || out <= 0 || B < 0 || ldx < in || ldy < out) return fail(c, FFH_ERR_BAD_ARG, "linear_fwd: bad dims");
  if (act != FFH_AC_MODE_NONE && act != FFH_AC_MODE_RELU && act != FFH_AC_MODE_SIGMOID && act != FFH_AC_MODE_GELU)
    return fail(c, FFH_ERR_UNSUPPORTED, "linear_fwd: activation");
  float* wt = (float*)malloc(sizeof(float) * (size_t)in * (size_t)out);   /* [in][out] */
  if (!wt) return fail(c, FFH_ERR_NOMEM, "oom");
  const int bf = use_bf16(c, in, out);      /* tensor-op mode: both operands rounded to bfloat16, fp32 products and sums */
  for (int o = 0; o < out; o++) for (int i = 0; i < in; i++) wt[(size_t)i * out + o] = bf ? bf16_round(w[(size_t)o * in + i]) : w[(size_t)o * in + i];
  /* blocks of LIN_BB samples share each weight row while it is in the cache (the weight matrix is streamed once per block, not once per
   * sample); an output element is still ONE i-ascending fma chain from 0: the same bits as the sample-by-sample loop */
  enum { LIN_BB = 8 };
#pragma omp parallel for schedule(static)
  for (int64_t b0 = 0; b0 < B; b0 += LIN_BB) {
    const int nb = (int)((B - b0) < LIN_BB ? (B - b0) : LIN_BB);
    for (int r = 0; r < nb; r++) { float* yr = y + (b0 + r) * ldy; for (int o = 0; o < out; o++) yr[o] = 0.0f; }
    for (int i = 0; i < in; i++) {
      const float* wr = wt + (size_t)i * out;
      for (int r = 0; r < nb; r++) {
        float* yr = y + (b0 + r) * ldy;
        const float xv = x[(b0 + r) * ldx + i];
        const float xi = bf ? bf16_round(xv) : xv;
        for (int o = 0; o < out; o++) yr[o] = fmaf(xi, wr[o], yr[o]);
      }
    }
    for (int r = 0; r < nb; r++) {
      float* yr = y + (b0 + r) * ldy;
      if (bias) for (int o = 0; o < out; o++) yr[o] = yr[o] + bias[o];
      if (act != FFH_AC_MODE_NONE) for (int o = 0; o < out; o++) yr[o] = act_fwd(yr[o], act);
    }
  }
  free(wt);
  return FFH_OK;
}

/* Linear::backward_kernel [ref: src/ops/linear.cu:610-660]:
 *   reluBackward [ref: src/runtime/cuda_helper.cu:71-78] / sigmoid_backward [ref: src/ops/linear.cu:600-607] in place on dy
 *   cublasSgemm(N,T, in,out,B) alpha=beta=1 : dw[o][i] += sum_b x[b][i] dy[b][o]
 *   cublasSgemv(N, out,B)      alpha=beta=1 : db[o]    += sum_b dy[b][o]
 *   cublasSgemm(N,N, in,B,out) alpha=beta=1 : dx[b][i] += sum_o w[o][i] dy[b][o]
 * Each product is an ascending FMA chain from 0 over the reduced index, then one add into
 * the accumulated buffer (C = 1*AB + 1*C). */
static float act_grad(float d, float yo, int act) {
  if (act == FFH_AC_MODE_RELU) return (yo > 0.0f) ? d : 0.0f;                 /* reluBackward [ref: src/runtime/cuda_helper.cu:71-78] */
  if (act == FFH_AC_MODE_SIGMOID) return d * yo * (1 - yo);                   /* sigmoid_backward [ref: src/ops/linear.cu:600-607] */
  return d;
}

/* parts of Linear::backward_kernel; mask_on_load: dy is read through act'(y) without being modified */
static int linear_bwd_parts(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx,
                            const float* y, int64_t ldy, float* dy, int64_t lddy,
                            const float* w, float* dw, float* db, int in, int out, int64_t B, int act,
                            int act_inplace, int do_dw, int do_db, int do_dx, int mask_on_load, int mask_by_x) {
  if (in <= 0 || out <= 0 || B < 0 || ldx < in || ldy < out || lddy < out || (dx && lddx < in))
    return fail(c, FFH_ERR_BAD_ARG, "linear_bwd: bad dims");
  if (act != FFH_AC_MODE_NONE && act != FFH_AC_MODE_RELU && act != FFH_AC_MODE_SIGMOID)
    return fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd: activation");
  if (act_inplace && act != FFH_AC_MODE_NONE) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++)
      for (int o = 0; o < out; o++)
        dy[b * lddy + o] = act_grad(dy[b * lddy + o], y[b * ldy + o], act);
  }
  const int m = mask_on_load ? act : FFH_AC_MODE_NONE;
  const int bf = use_bf16(c, in, out);      /* tensor-op mode: GEMM operands rounded to bfloat16 (db stays an fp32 sum of fp32 values) */
  if (do_dw) {
    /* blocks of LIN_OB output rows share each sample's x row while it is in the cache (x is streamed once per block of rows, not once per
     * row); an element of dW is still ONE b-ascending fma chain from 0, then one add */
    enum { LIN_OB = 8 };
#pragma omp parallel
    {
      float* tmp = (float*)malloc(sizeof(float) * (size_t)in * LIN_OB);
#pragma omp for schedule(static)
      for (int o0 = 0; o0 < out; o0 += LIN_OB) {
        const int no = (out - o0) < LIN_OB ? (out - o0) : LIN_OB;
        for (int i = 0; i < in * no; i++) tmp[i] = 0.0f;
        for (int64_t b = 0; b < B; b++) {
          const float* xr = x + b * ldx;
          for (int r = 0; r < no; r++) {
            float d = act_grad(dy[b * lddy + o0 + r], y[b * ldy + o0 + r], m);
            float* t = tmp + (size_t)r * in;
            if (bf) { d = bf16_round(d); for (int i = 0; i < in; i++) t[i] = fmaf(d, bf16_round(xr[i]), t[i]); }
            else for (int i = 0; i < in; i++) t[i] = fmaf(d, xr[i], t[i]);
          }
        }
        for (int r = 0; r < no; r++) for (int i = 0; i < in; i++) dw[(size_t)(o0 + r) * in + i] += tmp[(size_t)r * in + i];
      }
      free(tmp);
    }
  }
  if (do_db && db) {
    for (int o = 0; o < out; o++) {
      float acc = 0.0f;
      for (int64_t b = 0; b < B; b++) acc = fmaf(act_grad(dy[b * lddy + o], y[b * ldy + o], m), 1.0f, acc);
      db[o] += acc;
    }
  }
  if (do_dx && dx) {
    /* blocks of LIN_XB samples share each weight row (as the forward); an element of dX is still ONE o-ascending fma chain from 0, then one add */
    enum { LIN_XB = 8 };
#pragma omp parallel
    {
      float* tmp = (float*)malloc(sizeof(float) * (size_t)in * LIN_XB);
#pragma omp for schedule(static)
      for (int64_t b0 = 0; b0 < B; b0 += LIN_XB) {
        const int nb = (int)((B - b0) < LIN_XB ? (B - b0) : LIN_XB);
        for (int i = 0; i < in * nb; i++) tmp[i] = 0.0f;
        for (int o = 0; o < out; o++) {
          const float* wr = w + (size_t)o * in;
          for (int r = 0; r < nb; r++) {
            const int64_t b = b0 + r;
            float d = act_grad(dy[b * lddy + o], y[b * ldy + o], m);
            float* t = tmp + (size_t)r * in;
            if (bf) { d = bf16_round(d); for (int i = 0; i < in; i++) t[i] = fmaf(d, bf16_round(wr[i]), t[i]); }
            else for (int i = 0; i < in; i++) t[i] = fmaf(d, wr[i], t[i]);
          }
        }
        /* DX_MASK_BY_X: reluBackward of the layer below, applied to what this layer hands down */
        for (int r = 0; r < nb; r++) {
          const int64_t b = b0 + r;
          for (int i = 0; i < in; i++) dx[b * lddx + i] += (mask_by_x && !(x[b * ldx + i] > 0.0f)) ? 0.0f : tmp[(size_t)r * in + i];
        }
      }
      free(tmp);
    }
  }
  return FFH_OK;
}

int ffh_linear_bwd(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx,
                   const float* y, int64_t ldy, float* dy, int64_t lddy,
                   const float* w, float* dw, float* db,
                   int in, int out, int64_t B, int act, ffh_stream s) {
  (void)s;
  return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, act, 1, 1, 1, 1, 0, 0);
}

/* events order nothing on the host (every call has completed when it returns) */
int ffh_event_record_with_next_linear_bwd(ffh_ctx* c, ffh_event e) { (void)c; (void)e; return FFH_OK; }

/* no streams on the host: nothing ever runs on a second one */
int ffh_second_stream_used(ffh_ctx* c, int clear) { (void)c; (void)clear; return 0; }

/* same arithmetic; streams mean nothing on the host.  DX_OVERWRITE: dx is zeroed here, then accumulated */
static int linear_bwd_ex_plain(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx,
                               const float* y, int64_t ldy, float* dy, int64_t lddy,
                               const float* w, float* dw, float* db, int in, int out, int64_t B, int act, int flags);

/* the data gradient of the next call goes where a Concat backward would copy it (include/ff_hip.h); on the host the map is
 * always taken when the call qualifies */
int ffh_linear_bwd_set_dx_scatter(ffh_ctx* c, const ffh_col_dest* map, int ncols, ffh_event attach_if_used) {
  (void)attach_if_used;
  if (!c || !map || ncols <= 0) return FFH_ERR_BAD_ARG;
  c->scatter_map = map; c->scatter_ncols = ncols; c->scatter_used = 0;
  return FFH_OK;
}
int ffh_linear_dx_scatter_used(ffh_ctx* c) { return c ? c->scatter_used : 0; }

/* the lower layer's bias gradient as the column sums of the data gradient the next call stores (include/ff_hip.h); on the host the
 * request is always taken when the call qualifies (DX_OVERWRITE, in_dim == ncols, a data gradient is produced, no column map) */
int ffh_linear_bwd_set_dx_colsum(ffh_ctx* c, float* colsum, int ncols) {
  if (!c || !colsum || ncols <= 0) return FFH_ERR_BAD_ARG;
  c->colsum_dst = colsum; c->colsum_ncols = ncols; c->colsum_used = 0;
  return FFH_OK;
}
int ffh_linear_dx_colsum_used(ffh_ctx* c) { return c ? c->colsum_used : 0; }

int ffh_linear_bwd_ex(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx,
                      const float* y, int64_t ldy, float* dy, int64_t lddy,
                      const float* w, float* dw, float* db, int in, int out, int64_t B, int act,
                      int flags, ffh_stream s, ffh_stream s_dw) {
  (void)s; (void)s_dw;
  const ffh_col_dest* map = c ? c->scatter_map : NULL;
  const int take = map && c->scatter_ncols == in && (flags & FFH_LINEAR_DX_OVERWRITE) && !(flags & FFH_LINEAR_ONLY_DW) && dx;
  float* colsum = c ? c->colsum_dst : NULL;
  const int take_cs = colsum && c->colsum_ncols == in && (flags & FFH_LINEAR_DX_OVERWRITE) && !(flags & FFH_LINEAR_ONLY_DW) && dx && !take;
  if (c) { c->scatter_map = NULL; c->scatter_used = 0; c->colsum_dst = NULL; c->colsum_used = 0; }
  const int rc = linear_bwd_ex_plain(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, act, flags);
  if (rc == FFH_OK && take_cs) {       /* ascending chain over the rows, as the db of linear_bwd_parts */
    for (int n = 0; n < in; n++) {
      float acc = 0.0f;
      for (int64_t b = 0; b < B; b++) acc = fmaf(dx[b * lddx + n], 1.0f, acc);
      colsum[n] += acc;
    }
    c->colsum_used = 1;
  }
  if (rc == FFH_OK && take) {
    for (int64_t b = 0; b < B; b++)
      for (int n = 0; n < in; n++) map[n].base[b * map[n].ld] = dx[b * lddx + n];
    c->scatter_used = 1;
  }
  return rc;
}

static int linear_bwd_ex_plain(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx,
                               const float* y, int64_t ldy, float* dy, int64_t lddy,
                               const float* w, float* dw, float* db, int in, int out, int64_t B, int act, int flags) {
  const int only_dx = flags & FFH_LINEAR_ONLY_DX, only_dw = flags & FFH_LINEAR_ONLY_DW;
  if (only_dx && only_dw) return fail(c, FFH_ERR_BAD_ARG, "linear_bwd_ex: ONLY_DX and ONLY_DW are exclusive");
  if ((flags & FFH_LINEAR_DX_OVERWRITE) && !only_dw && dx && in > 0 && lddx >= in)
    for (int64_t b = 0; b < B; b++) memset(dx + b * lddx, 0, sizeof(float) * (size_t)in);
  const int mbx = (flags & FFH_LINEAR_DX_MASK_BY_X) ? 1 : 0;
  if (flags & FFH_LINEAR_DY_PREMASKED) {
    /* the activation derivative was applied by the producer of dy: this layer is linear in dy */
    if (only_dx) return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, FFH_AC_MODE_NONE, 0, 0, 0, 1, 0, mbx);
    if (only_dw) return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, FFH_AC_MODE_NONE, 0, 1, 1, 0, 0, 0);
    return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, FFH_AC_MODE_NONE, 0, 1, 1, 1, 0, mbx);
  }
  const int sig = act == FFH_AC_MODE_SIGMOID;
  if (only_dx)   /* sigmoid: in-place pass + db here; relu: dy untouched, masked while read */
    return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, act, sig, 0, sig, 1, !sig, mbx);
  if (only_dw)   /* relu: mask written back in place; sigmoid: dy already transformed, db already done */
    return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, sig ? FFH_AC_MODE_NONE : act, !sig, 1, !sig, 0, 0, 0);
  return linear_bwd_parts(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, act, 1, 1, 1, 1, 0, mbx);
}

/* the two calls it stands for, in order (include/ff_hip.h) */
int ffh_linear_bwd_mse(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy, float* dy, int64_t lddy,
                       const float* w, float* dw, float* db, int in, int out, int64_t B, int act, int flags,
                       const float* label, float scale, ffh_perf_metrics* perf, int metrics_flags, ffh_stream s) {
  if (flags & (FFH_LINEAR_ONLY_DX | FFH_LINEAR_ONLY_DW | FFH_LINEAR_DY_PREMASKED)) return fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: split / premasked forms");
  if (out > 4 || in > 1024 || ldy != out || lddy != out) return fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: not a one-launch layer");
  const int rc = ffh_mse_bwd_metrics(c, dy, y, label, perf, B, out, scale, metrics_flags, s);
  if (rc != FFH_OK) return rc;
  return ffh_linear_bwd_ex(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, B, act, flags, s, s);
}

int ffh_linear_pair_fwd(ffh_ctx* c, const float* x_l, int64_t ldx_l, const float* w_l, const float* b_l, int in_l, int act_l,
                        float* y_l, int64_t ldy_l, int mid, const float* w_u, const float* b_u, int out_u, int act_u,
                        float* y_u, int64_t ldy_u, int64_t B, ffh_stream s) {
  const int ks = (mid == 32 || mid == 64) ? 8 / (mid / 32) : 1;
  if (out_u > 16 || (mid != 32 && mid != 64) || in_l % (32 * ks) != 0) return fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_fwd: shapes");
  const int rc = ffh_linear_fwd(c, x_l, ldx_l, y_l, ldy_l, w_l, b_l, in_l, mid, B, act_l, s);
  if (rc != FFH_OK) return rc;
  return ffh_linear_fwd(c, y_l, ldy_l, y_u, ldy_u, w_u, b_u, mid, out_u, B, act_u, s);
}

/* the two calls it stands for (include/ff_hip.h) */
int ffh_linear_pair_bwd(ffh_ctx* c, const float* x_u, int64_t ldx_u, const float* y_u, int64_t ldy_u, float* dy_u, int64_t lddy_u,
                        const float* w_u, float* dw_u, float* db_u, int in_u, int out_u, int act_u, int flags_u,
                        const float* x_l, int64_t ldx_l, float* dx_l, int64_t lddx_l, float* dy_l, int64_t lddy_l, const float* w_l,
                        int in_l, int act_l, int flags_l, int64_t B, ffh_stream s) {
  if ((flags_u & ~FFH_LINEAR_DY_PREMASKED) || (flags_l & ~(FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X)))
    return fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: flags");
  if (out_u > 16 || (in_u != 32 && in_u != 64) || in_l % 32 != 0 || (act_l != FFH_AC_MODE_RELU && act_l != FFH_AC_MODE_NONE))
    return fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: shapes");
  int rc = ffh_linear_bwd_ex(c, x_u, ldx_u, dy_l, lddy_l, y_u, ldy_u, dy_u, lddy_u, w_u, dw_u, db_u, in_u, out_u, B, act_u,
                             flags_u | FFH_LINEAR_DX_OVERWRITE | (act_l == FFH_AC_MODE_RELU ? FFH_LINEAR_DX_MASK_BY_X : 0), s, s);
  if (rc != FFH_OK) return rc;
  return ffh_linear_bwd_ex(c, x_l, ldx_l, dx_l, lddx_l, x_u, ldx_u, dy_l, lddy_l, w_l, NULL, NULL, in_l, in_u, B, act_l,
                           flags_l | FFH_LINEAR_ONLY_DX | FFH_LINEAR_DY_PREMASKED, s, s);
}

/* A chain of narrow Linear layers: the per-layer calls it stands for, in order (include/ff_hip.h).  The per-layer functions take
 * w as [out][in] dense; a layer with ldw > in goes through a compact copy. */
static int chain_check(ffh_ctx* c, const ffh_chain_layer* ls, int n, int64_t B, const char* who) {
  (void)who;
  if (!ls || n < 1 || n > FFH_CHAIN_MAX_LAYERS || B < 0) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain: layer count / batch");
  for (int l = 0; l < n; l++) {
    if (ls[l].in_dim < 1 || ls[l].out_dim < 1 || ls[l].in_dim > FFH_CHAIN_MAX_WIDTH || ls[l].out_dim > FFH_CHAIN_MAX_WIDTH) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain: layer width (1 .. 512)");
    if (ls[l].ldw < ls[l].in_dim || ls[l].ldy < ls[l].out_dim || !ls[l].w || !ls[l].y) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain: leading dimension / null pointer");
    if (l > 0 && ls[l].in_dim != ls[l - 1].out_dim) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain: widths do not chain");
  }
  return FFH_OK;
}
static float* chain_compact(const float* w, int out, int in, int ldw) {
  float* t = (float*)malloc(sizeof(float) * (size_t)out * (size_t)in);
  if (t) for (int o = 0; o < out; o++) memcpy(t + (size_t)o * in, w + (size_t)o * ldw, sizeof(float) * (size_t)in);
  return t;
}
/* the chain entries' math-mode rule (csrc/mlp_chain.hip, chain_math_mode_ok): exact mode, or the split mode proper with every layer below its flop rule */
static int chain_math_mode_ok(const ffh_ctx* c, const ffh_chain_layer* ls, int n, int64_t B) {
  if (!c || c->math_mode == FFH_MATH_DEFAULT) return 1;
  if (c->math_mode != FFH_MATH_FP32_SPLIT_BF16X3) return 0;
  for (int l = 0; l < n; l++)
    if (ls[l].in_dim >= FFH_BF16_MIN_DIM && ls[l].out_dim >= FFH_BF16_MIN_DIM && 2.0 * (double)B * (double)ls[l].in_dim * (double)ls[l].out_dim >= FFH_BF16X3_MIN_FLOP) return 0;
  return 1;
}
int ffh_mlp_chain_fwd(ffh_ctx* c, const float* x, int64_t ldx, const ffh_chain_layer* ls, int n, int64_t B, ffh_stream s) {
  int rc = chain_check(c, ls, n, B, "fwd");
  if (rc != FFH_OK) return rc;
  if (!chain_math_mode_ok(c, ls, n, B)) return fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: layers of the exact-fp32 kernels only (math mode)");
  for (int l = 0; l < n && rc == FFH_OK; l++) {
    const ffh_chain_layer* L = &ls[l];
    const float* xin = l == 0 ? x : ls[l - 1].y;
    const int64_t ldi = l == 0 ? ldx : ls[l - 1].ldy;
    float* wc = L->ldw != L->in_dim ? chain_compact(L->w, L->out_dim, L->in_dim, L->ldw) : NULL;
    if (L->ldw != L->in_dim && !wc) return fail(c, FFH_ERR_NOMEM, "oom");
    rc = ffh_linear_fwd(c, xin, ldi, L->y, L->ldy, wc ? wc : L->w, L->bias, L->in_dim, L->out_dim, B, L->activation, s);
    free(wc);
  }
  return rc;
}
int ffh_mlp_chain_bwd(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const ffh_chain_layer* ls, int n, int64_t B, int flags,
                      ffh_stream s) {
  int rc = chain_check(c, ls, n, B, "bwd");
  if (rc != FFH_OK) return rc;
  if (flags & ~(FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X | FFH_LINEAR_DY_PREMASKED)) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain_bwd: flags");
  for (int l = 0; l < n; l++)
    if (!ls[l].dy || !ls[l].dw || ls[l].lddy < ls[l].out_dim) return fail(c, FFH_ERR_BAD_ARG, "mlp_chain_bwd: null pointer / leading dimension");
  if (!chain_math_mode_ok(c, ls, n, B)) return fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: layers of the exact-fp32 kernels only (math mode)");
  for (int l = 0; l + 1 < n; l++)
    if (ls[l].activation != FFH_AC_MODE_NONE && ls[l].activation != FFH_AC_MODE_RELU) return fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: inner layers NONE or RELU");
  for (int l = n - 1; l >= 0 && rc == FFH_OK; l--) {
    const ffh_chain_layer* L = &ls[l];
    const float* xin = l == 0 ? x : ls[l - 1].y;
    const int64_t ldi = l == 0 ? ldx : ls[l - 1].ldy;
    float* dxl = l == 0 ? dx : ls[l - 1].dy;
    const int64_t lddxl = l == 0 ? lddx : ls[l - 1].lddy;
    int f = 0;
    if (l == n - 1) f |= flags & FFH_LINEAR_DY_PREMASKED;
    else if (L->activation == FFH_AC_MODE_RELU) f |= FFH_LINEAR_DY_PREMASKED;
    if (l == 0) f |= flags & (FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X);
    else f |= FFH_LINEAR_DX_OVERWRITE | (ls[l - 1].activation == FFH_AC_MODE_RELU ? FFH_LINEAR_DX_MASK_BY_X : 0);
    float *wc = NULL, *dwc = NULL;
    if (L->ldw != L->in_dim) {
      wc = chain_compact(L->w, L->out_dim, L->in_dim, L->ldw);
      dwc = chain_compact(L->dw, L->out_dim, L->in_dim, L->ldw);
      if (!wc || !dwc) { free(wc); free(dwc); return fail(c, FFH_ERR_NOMEM, "oom"); }
    }
    rc = ffh_linear_bwd_ex(c, xin, ldi, dxl, lddxl, L->y, L->ldy, L->dy, L->lddy, wc ? wc : L->w, dwc ? dwc : L->dw, L->db, L->in_dim, L->out_dim, B,
                           L->activation, f, s, s);
    if (dwc) for (int o = 0; o < L->out_dim; o++) memcpy(L->dw + (size_t)o * L->ldw, dwc + (size_t)o * L->in_dim, sizeof(float) * (size_t)L->in_dim);
    free(wc); free(dwc);
  }
  return rc;
}

/* ------------------------------------------------------------------ */
/* Concat                                                             */
/* ------------------------------------------------------------------ */
/* Concat::forward_kernel [ref: src/ops/concat.cu:211-249] + copy_with_stride
 * [ref: src/runtime/cuda_helper.cu:128-144] */
int ffh_concat_fwd(ffh_ctx* c, float* out, int64_t out_blk, const float* const* ins,
                   const int64_t* in_blk, const int64_t* in_ld, int n, int64_t nblk, ffh_stream s) {
  (void)s;
  if (n < 0 || n > FFH_MAX_CONCAT_INPUTS || nblk < 0) return fail(c, FFH_ERR_BAD_ARG, "concat_fwd: bad dims");
  int64_t off = 0;
  for (int i = 0; i < n; i++) {
    const int64_t ld = in_ld ? in_ld[i] : in_blk[i];
    if (in_blk[i] < 0 || ld < in_blk[i] || off + in_blk[i] > out_blk) return fail(c, FFH_ERR_BAD_ARG, "concat_fwd: widths");
    if (!(ins[i] == out + off && ld == out_blk))
      for (int64_t b = 0; b < nblk; b++)
        for (int64_t e = 0; e < in_blk[i]; e++)
          out[b * out_blk + off + e] = ins[i][b * ld + e];
    off += in_blk[i];
  }
  return FFH_OK;
}
/* Concat::backward_kernel [ref: src/ops/concat.cu:325-360] + add_with_stride
 * [ref: src/runtime/cuda_helper.cu:110-126] */
int ffh_concat_bwd_ex(ffh_ctx* c, const float* og, int64_t out_blk, float* const* igs,
                      const int64_t* in_blk, const int64_t* in_ld, int n, int64_t nblk, int flags, ffh_stream s) {
  (void)s;
  if (n < 0 || n > FFH_MAX_CONCAT_INPUTS || nblk < 0) return fail(c, FFH_ERR_BAD_ARG, "concat_bwd: bad dims");
  if (flags & ~FFH_CONCAT_BWD_OVERWRITE) return fail(c, FFH_ERR_BAD_ARG, "concat_bwd_ex: unknown flags");
  const int ow = flags & FFH_CONCAT_BWD_OVERWRITE;
  int64_t off = 0;
  for (int i = 0; i < n; i++) {
    const int64_t ld = in_ld ? in_ld[i] : in_blk[i];
    if (in_blk[i] < 0 || ld < in_blk[i] || off + in_blk[i] > out_blk) return fail(c, FFH_ERR_BAD_ARG, "concat_bwd: widths");
    if (igs[i] && !(igs[i] == og + off && ld == out_blk))
      for (int64_t b = 0; b < nblk; b++)
        for (int64_t e = 0; e < in_blk[i]; e++) {
          const float g = og[b * out_blk + off + e];
          igs[i][b * ld + e] = ow ? g : igs[i][b * ld + e] + g;   /* add_with_stride [ref: src/runtime/cuda_helper.cu:110-126] / plain store */
        }
    off += in_blk[i];
  }
  return FFH_OK;
}

int ffh_concat_bwd(ffh_ctx* c, const float* og, int64_t out_blk, float* const* igs,
                   const int64_t* in_blk, const int64_t* in_ld, int n, int64_t nblk, ffh_stream s) {
  return ffh_concat_bwd_ex(c, og, out_blk, igs, in_blk, in_ld, n, nblk, 0, s);
}

/* ------------------------------------------------------------------ */
/* BatchMatmul                                                        */
/* ------------------------------------------------------------------ */
/* BatchMatmul::forward_kernel [ref: src/ops/batch_matmul.cu:194-244]: strides are taken
 * from the FULL n,k,m (:212-215) before seq_length shrinks k / n / m (:216-236); then
 * cublasSgemmStridedBatched(N,N, m,n,k), beta 0: o[b][r][c] = sum_q a[b][r][q] b[b][q][c]. */
int ffh_bmm_fwd(ffh_ctx* c, float* o, const float* a, const float* b,
                int m, int n, int k, int64_t batch, int asd, int bsd, int seq, ffh_stream s) {
  (void)s;
  if (m <= 0 || n <= 0 || k <= 0 || batch < 0) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: bad dims");
  const int lda = k, ldb = m, ldo = m;
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
  if (asd == 0 && seq >= 0) { if (seq > k || bsd != 1) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: seq_length"); k = seq; }
  else if (asd == 1 && seq >= 0) { if (seq > n) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: seq_length"); n = seq; }
  else if (!(asd < 0 || seq < 0)) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: a_seq_length_dim");
  if (bsd == 0 && seq >= 0) { if (seq > m) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: seq_length"); m = seq; }
  else if (bsd == 1 && seq >= 0) { if (asd != 0 || k != seq) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: seq_length"); }
  else if (!(bsd < 0 || seq < 0)) return fail(c, FFH_ERR_BAD_ARG, "bmm_fwd: b_seq_length_dim");
#pragma omp parallel for schedule(static)
  for (int64_t bi = 0; bi < batch; bi++) {
    const float* A = a + bi * sa; const float* Bm = b + bi * sb; float* O = o + bi * so;
    for (int r = 0; r < n; r++) {
      for (int cc = 0; cc < m; cc++) O[(size_t)r * ldo + cc] = 0.0f;
      for (int q = 0; q < k; q++) {
        const float av = A[(size_t)r * lda + q];
        for (int cc = 0; cc < m; cc++) O[(size_t)r * ldo + cc] = fmaf(av, Bm[(size_t)q * ldb + cc], O[(size_t)r * ldo + cc]);
      }
    }
  }
  return FFH_OK;
}
/* BatchMatmul::backward_kernel [ref: src/ops/batch_matmul.cu:375-400]:
 *   a_grad[b][r][q] += sum_c o_grad[b][r][c] * B[b][q][c]
 *   b_grad[b][q][c] += sum_r A[b][r][q] * o_grad[b][r][c] */
int ffh_bmm_bwd(ffh_ctx* c, const float* og, const float* a, float* ag, const float* b, float* bg,
                int m, int n, int k, int64_t batch, ffh_stream s) {
  (void)s;
  if (m <= 0 || n <= 0 || k <= 0 || batch < 0) return fail(c, FFH_ERR_BAD_ARG, "bmm_bwd: bad dims");
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
#pragma omp parallel for schedule(static)
  for (int64_t bi = 0; bi < batch; bi++) {
    const float* A = a + bi * sa; const float* Bm = b + bi * sb; const float* G = og + bi * so;
    float* AG = ag + bi * sa; float* BG = bg + bi * sb;
    for (int r = 0; r < n; r++)
      for (int q = 0; q < k; q++) {
        float acc = 0.0f;
        for (int cc = 0; cc < m; cc++) acc = fmaf(G[(size_t)r * m + cc], Bm[(size_t)q * m + cc], acc);
        AG[(size_t)r * k + q] += acc;
      }
    for (int q = 0; q < k; q++)
      for (int cc = 0; cc < m; cc++) {
        float acc = 0.0f;
        for (int r = 0; r < n; r++) acc = fmaf(A[(size_t)r * k + q], G[(size_t)r * m + cc], acc);
        BG[(size_t)q * m + cc] += acc;
      }
  }
  return FFH_OK;
}

/* ------------------------------------------------------------------ */
/* Transpose                                                          */
/* ------------------------------------------------------------------ */
/* transpose_simple_kernel [ref: src/ops/transpose.cu:195-251]: every output index is decomposed by the output
 * strides and re-assembled with the permuted input strides.  dims/perm here are in natural order. */
static int transpose_impl(ffh_ctx* c, float* dst, const float* src, int nd, const int64_t* in_dims, const int* perm, int backward) {
  if (nd < 1 || nd > 4) return fail(c, FFH_ERR_BAD_ARG, "transpose: ndim must be 1..4");
  int seen[4] = {0, 0, 0, 0};
  int64_t od[4], is[4], vol = 1;
  for (int i = 0; i < nd; i++) {
    if (perm[i] < 0 || perm[i] >= nd || seen[perm[i]] || in_dims[i] <= 0) return fail(c, FFH_ERR_BAD_ARG, "transpose: bad perm/dims");
    seen[perm[i]] = 1;
  }
  for (int i = nd - 1; i >= 0; i--) { is[i] = (i == nd - 1) ? 1 : is[i + 1] * in_dims[i + 1]; vol *= in_dims[i]; }
  for (int i = 0; i < nd; i++) od[i] = in_dims[perm[i]];
  for (int64_t o = 0; o < vol; o++) {
    int64_t t = o, ii = 0;
    for (int i = nd - 1; i >= 0; i--) { const int64_t q = t % od[i]; t /= od[i]; ii += q * is[perm[i]]; }
    if (backward) dst[ii] += src[o];      /* in_grad[ii] += out_grad[o] */
    else dst[o] = src[ii];
  }
  return FFH_OK;
}
int ffh_transpose_fwd(ffh_ctx* c, float* out, const float* in, int nd, const int64_t* in_dims, const int* perm, ffh_stream s) {
  (void)s; return transpose_impl(c, out, in, nd, in_dims, perm, 0);
}
int ffh_transpose_bwd(ffh_ctx* c, float* in_grad, const float* out_grad, int nd, const int64_t* in_dims, const int* perm, ffh_stream s) {
  (void)s; return transpose_impl(c, in_grad, out_grad, nd, in_dims, perm, 1);
}

/* ------------------------------------------------------------------ */
/* Loss / metrics / optimizer                                         */
/* ------------------------------------------------------------------ */
/* mean_squared_error_avg_loss_backward [ref: src/loss_functions/loss_functions.cu:65-76]
 * then scale_kernel(ptr, n, 0, scale) [ref: :160-166; src/runtime/cuda_helper.cu:33-40]:
 * ptr = (scale - 0)*ptr + 0  == fl(scale * fl(logit - label)) */
/* strict lower triangle of the pairwise-dot matrix (no reference operator; MLPerf-DLRM / torch: Z[:, li, lj] with
 * tril_indices(n, n, -1), i.e. i ascending, j < i ascending) */
int ffh_tril_fwd(ffh_ctx* c, float* out, int64_t out_ld, const float* in, int64_t batch, int n, ffh_stream s) {
  (void)s;
  if (batch < 0 || n < 2 || n > 64 || out_ld < (int64_t)n * (n - 1) / 2 || (batch > 0 && (!out || !in))) return fail(c, FFH_ERR_BAD_ARG, "tril_fwd: bad args");
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; b++) {
    int64_t p = 0;
    for (int i = 1; i < n; i++)
      for (int j = 0; j < i; j++) out[b * out_ld + p++] = in[(b * n + i) * n + j];
  }
  return FFH_OK;
}
int ffh_tril_bwd(ffh_ctx* c, float* in_grad, const float* out_grad, int64_t grad_ld, int64_t batch, int n, ffh_stream s) {
  (void)s;
  if (batch < 0 || n < 2 || n > 64 || grad_ld < (int64_t)n * (n - 1) / 2 || (batch > 0 && (!in_grad || !out_grad))) return fail(c, FFH_ERR_BAD_ARG, "tril_bwd: bad args");
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; b++) {
    int64_t p = 0;
    for (int i = 1; i < n; i++)
      for (int j = 0; j < i; j++) in_grad[(b * n + i) * n + j] += out_grad[b * grad_ld + p++];
  }
  return FFH_OK;
}

/* the fused pairwise-dot interaction (include/ff_hip.h): plain loops, k ascending */
int ffh_dot_interaction_fwd(ffh_ctx* c, const float* z, int64_t ldz, float* out, int64_t ldo, int64_t batch, int nrows, int d, ffh_stream s) {
  (void)s;
  if (batch < 0 || nrows < 2 || nrows > 32 || d < 1 || ldz < (int64_t)nrows * d || ldo < d + (int64_t)nrows * (nrows - 1) / 2 ||
      (batch > 0 && (!z || !out)))
    return fail(c, FFH_ERR_BAD_ARG, "dot_interaction_fwd: bad args");
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; b++) {
    const float* zb = z + b * ldz;
    float* ob = out + b * ldo;
    for (int k = 0; k < d; k++) ob[k] = zb[k];
    int64_t p = d;
    for (int i = 1; i < nrows; i++)
      for (int j = 0; j < i; j++) {
        float acc = 0.0f;
        for (int k = 0; k < d; k++) acc = acc + zb[(int64_t)i * d + k] * zb[(int64_t)j * d + k];
        ob[p++] = acc;
      }
  }
  return FFH_OK;
}
int ffh_dot_interaction_bwd(ffh_ctx* c, const float* z, int64_t ldz, const float* og, int64_t ldg, float* zg, int64_t ldzg,
                            int64_t batch, int nrows, int d, int flags, ffh_stream s) {
  (void)s;
  if (batch < 0 || nrows < 2 || nrows > 32 || d < 1 || ldz < (int64_t)nrows * d || ldzg < (int64_t)nrows * d ||
      ldg < d + (int64_t)nrows * (nrows - 1) / 2 || (batch > 0 && (!z || !og || !zg)) || (flags & ~FFH_DOT_BWD_OVERWRITE))
    return fail(c, FFH_ERR_BAD_ARG, "dot_interaction_bwd: bad args");
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; b++) {
    const float* zb = z + b * ldz;
    const float* gb = og + b * ldg;
    float* db = zg + b * ldzg;
    for (int i = 0; i < nrows; i++)
      for (int k = 0; k < d; k++) {
        float acc = 0.0f;
        for (int j = 0; j < nrows; j++) {
          if (j == i) continue;
          const int hi = i > j ? i : j, lo = i > j ? j : i;
          acc = acc + gb[d + (int64_t)hi * (hi - 1) / 2 + lo] * zb[(int64_t)j * d + k];
        }
        if (i == 0) acc = acc + gb[k];
        db[(int64_t)i * d + k] = (flags & FFH_DOT_BWD_OVERWRITE) ? acc : db[(int64_t)i * d + k] + acc;
      }
  }
  return FFH_OK;
}

int ffh_mse_bwd(ffh_ctx* c, float* lg, const float* logit, const float* label, int64_t n, float scale, ffh_stream s) {
  (void)c; (void)s;
  for (int64_t i = 0; i < n; i++) {
    const float d = logit[i] - label[i];
    lg[i] = fmaf(scale - 0.0f, d, 0.0f);
  }
  return FFH_OK;
}

int ffh_metrics_update(ffh_ctx* c, const float* logits, const float* labels, ffh_perf_metrics* perf,
                       int64_t ns, int nc, int flags, ffh_stream s);
/* FFModel::backward's compute_metrics + loss backward [ref: src/runtime/model.cc:1443-1452] */
int ffh_mse_bwd_metrics(ffh_ctx* c, float* lg, const float* logit, const float* label, ffh_perf_metrics* perf,
                        int64_t ns, int nc, float scale, int flags, ffh_stream s) {
  int rc = ffh_metrics_update(c, logit, label, perf, ns, nc, flags, s);
  if (rc) return rc;
  return ffh_mse_bwd(c, lg, logit, label, ns * nc, scale, s);
}

/* update_metrics_label_kernel [ref: src/metrics_functions/metrics_functions.cu:108-173];
 * atomics arrive in undefined order in the reference, here b ascending. */
int ffh_metrics_update(ffh_ctx* c, const float* logits, const float* labels, ffh_perf_metrics* perf,
                       int64_t ns, int nc, int flags, ffh_stream s) {
  (void)s;
  if (nc <= 0 || ns < 0 || !perf) return fail(c, FFH_ERR_BAD_ARG, "metrics_update: bad dims");
  for (int64_t b = 0; b < ns; b++) {
    perf->train_all += 1;
    if (flags & 1) {
      if (nc == 1) { perf->train_all += 1; perf->train_correct += 1; }
      else {
        float max_val = 0.0f; int my = -1, tr = -1;
        for (int i = 0; i < nc; i++) {
          if (my == -1 || logits[b * nc + i] > max_val) { max_val = logits[b * nc + i]; my = i; }
          if (labels[b * nc + i] > 0.9f) tr = i;
        }
        if (tr == my) perf->train_correct += 1;
      }
    }
    if (flags & (2 | 4 | 8)) {
      float mse = 0.0f, mae = 0.0f;
      for (int i = 0; i < nc; i++) {
        const float diff = logits[b * nc + i] - labels[b * nc + i];
        mse = fmaf(diff, diff, mse);
        mae += fabsf(diff);
      }
      if (flags & 2) perf->mse_loss += mse;
      if (flags & 4) perf->rmse_loss += sqrtf(mse);
      if (flags & 8) perf->mae_loss += mae;
    }
  }
  return FFH_OK;
}

/* sgd_update [ref: src/runtime/optimizer_kernel.cu:23-41], with the multiply-adds
 * contracted the way nvcc's default -fmad=true contracts them.  FFH_OPT_ZERO_GRAD clears the
 * consumed gradient (what the next step's zero_grad [ref: src/runtime/model.cc:466-490] would do). */
int ffh_sgd_update_ex(ffh_ctx* c, float* w, float* g, float* v, int64_t n,
                      float lr, float wd, float mom, int nesterov, int flags, ffh_stream s) {
  (void)s;
  if (mom > 0.0f && !v) return fail(c, FFH_ERR_BAD_ARG, "sgd_update: momentum without V");
  if (flags & ~FFH_OPT_ZERO_GRAD) return fail(c, FFH_ERR_BAD_ARG, "sgd_update: unknown flags");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    float gt = fmaf(wd, w[i], g[i]);
    if (mom > 0.0f) {
      v[i] = fmaf(v[i], mom, gt);
      if (nesterov) gt = fmaf(mom, v[i], gt); else gt = v[i];
    }
    w[i] = fmaf(-lr, gt, w[i]);
    if (flags & FFH_OPT_ZERO_GRAD) g[i] = 0.0f;
  }
  return FFH_OK;
}

int ffh_sgd_update(ffh_ctx* c, float* w, const float* g, float* v, int64_t n,
                   float lr, float wd, float mom, int nesterov, ffh_stream s) {
  return ffh_sgd_update_ex(c, w, (float*)g, v, n, lr, wd, mom, nesterov, 0, s);
}

/* adam_update [ref: src/runtime/optimizer_kernel.cu:206-226], one statement of the reference per line;
 * the a*b+c forms are single fused multiply-adds (the canonical rounding of include/ff_hip.h), sqrtf and
 * the division are IEEE correctly rounded. */
int ffh_adam_update(ffh_ctx* c, float* w, float* g, float* m, float* v, int64_t n, float alpha_t, float beta1,
                    float beta2, float wd, float eps, int flags, ffh_stream s) {
  (void)s;
  if (n > 0 && (!w || !g || !m || !v)) return fail(c, FFH_ERR_BAD_ARG, "adam_update: bad args");
  if (flags & ~FFH_OPT_ZERO_GRAD) return fail(c, FFH_ERR_BAD_ARG, "adam_update: unknown flags");
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    const float gt = fmaf(wd, w[i], g[i]);                 /* gt = WGrad[i] + weight_decay * W[i]            :218 */
    const float mt = fmaf(beta1, m[i], omb1 * gt);         /* mt = beta1 * M[i] + (1 - beta1) * gt           :219 */
    const float vt = fmaf(beta2, v[i], (omb2 * gt) * gt);  /* vt = beta2 * V[i] + (1 - beta2) * gt * gt      :220 */
    m[i] = mt;
    v[i] = vt;
    w[i] = w[i] - (alpha_t * mt) / (sqrtf(vt) + eps);      /* W[i] -= alpha_t * mt / (sqrt(vt) + epsilon)    :223 */
    if (flags & FFH_OPT_ZERO_GRAD) g[i] = 0.0f;
  }
  return FFH_OK;
}

/* apply_add_with_scale [ref: src/runtime/cuda_helper.cu:99-108] */
/* dst[i] = the slices' i-th elements added in slice order (ffh_sum_slices_f32: the local step of the direct all-reduce) */
int ffh_sum_slices_f32(ffh_ctx* c, float* d, const float* src, int nslices, int64_t n, int64_t stride, ffh_stream s) {
  (void)s;
  if (!c || n < 0 || nslices < 1 || (nslices > 1 && stride < n) || ((!d || !src) && n)) return FFH_ERR_BAD_ARG;
  for (int64_t i = 0; i < n; i++) {
    float v = src[i];
    for (int q = 1; q < nslices; q++) v = v + src[(int64_t)q * stride + i];
    d[i] = v;
  }
  return FFH_OK;
}
int ffh_add_scaled(ffh_ctx* c, float* d, const float* src, int64_t n, float scale, ffh_stream s) {
  (void)c; (void)s;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) d[i] = fmaf(src[i], scale, d[i]);
  return FFH_OK;
}
