/* ff_hip.h -- C-ABI of the MI355X-native DLRM operator kernels.
 *
 * This is the drop-in boundary for the DLRM training hot path of
 * facebookresearch/DLRM-FlexFlow.  Every entry point replaces one of the
 * reference's static `X::forward_kernel / backward_kernel` members (the tier
 * that its Legion task wrappers, FusedOp and the simulator all call with raw
 * device pointers, integer dims and a stream -- SURVEY.md section 8b).  The
 * reference interface each function replaces is cited as
 * `[ref: file:line]` (paths relative to the reference checkout).
 *
 * Conventions
 *  - POD arguments only: raw device pointers, sizes, enum values as int
 *    (values identical to the reference's include/ffconst.h:4-57), a stream
 *    as `void*` (a hipStream_t).  No C++/torch types cross this boundary.
 *  - Every tensor is a flat row-major buffer with the batch outermost
 *    (Legion dims reversed, [ref: src/runtime/model.cc:865-868]).  Where a
 *    function takes a leading dimension (`ld*`, in elements) the operand may
 *    be a column slice of a wider buffer (e.g. an embedding output living
 *    inside the concat buffer); `ld == row width` gives the reference's dense
 *    layout.
 *  - All compute entry points are asynchronous on the caller's stream, never
 *    synchronise, never allocate; scratch comes from the workspace the caller
 *    attached to the ctx (the reference's FFHandler.workSpace,
 *    [ref: include/config.h:75-84]).
 *  - Return value: 0 (FFH_OK) or a negative FFH_ERR_*; the text of the last
 *    error on a ctx is available from ffh_last_error_string().  The library
 *    never exits or asserts across the ABI (the reference's
 *    checkCUDA/assert abort behaviour, [ref: include/cuda_helper.h:6-47], is
 *    re-created by the C++ FFModel shim on a non-zero return).
 *  - A ctx is bound to one device and is not thread-safe; different ctxs are
 *    independent.
 *
 * Two libraries export this ABI:
 *   dlrm_flexflow_amd/csrc  -> libffhip.so     the product (hand-written gfx950 HIP)
 *   oracle/                 -> libffh_oracle.so test-only CPU restatement of the
 *                                               reference arithmetic ("device"
 *                                               pointers are host pointers)
 * Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the second one.
 */
#ifndef FF_HIP_H_
#define FF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFH_ABI_VERSION 14   /* 14: ffh_ctx_bf16x3_mirror_set, ffh_convert_f32_to_bf16x3 (three-plane images for the fp32-accurate split mode), ffh_sum_slices_f32 (direct all-reduce); 13: ffh_embedding_last_route; 12: ffh_mlp_chain_fwd / _bwd (a chain of narrow Linear layers as three launches), ffh_ctx_reserve_scratch; 11: ffh_stream_create_with_priority; 10: ffh_linear_bwd_set_dx_colsum, ffh_linear_dx_colsum_used; 9: ffh_ctx_set_dw_cu_reserve; 8: ffh_embedding_bwd_opt_fused_multi / _apply_multi (sparse momentum-SGD / Adam on the sorted segments); 7: ffh_embedding_bwd_sort_multi, ffh_embedding_bwd_sgd_apply_multi; 6: ffh_ctx_bf16_mirror_set, ffh_convert_f32_to_bf16; 5: ffh_ctx_default, ffh_linear_last_route; 4: ffh_ctx_set_math_mode, ffh_ctx_set_deterministic; 2: optimizer / linear / concat *_ex entry points, ffh_adam_update, ffh_second_stream_used; 3: ffh_embedding_localize_rows, ffh_tril_*, ffh_dot_interaction_*, ffh_linear_bwd_mse, ffh_linear_pair_fwd / _bwd, ffh_linear_bwd_set_dx_scatter */

/* status codes */
#define FFH_OK               0
#define FFH_ERR_BAD_ARG     (-1)
#define FFH_ERR_HIP         (-2)
#define FFH_ERR_UNSUPPORTED (-3)
#define FFH_ERR_WORKSPACE   (-4)
#define FFH_ERR_NOMEM       (-5)

/* enum values, identical to [ref: include/ffconst.h:4-57] */
#define FFH_AC_MODE_NONE     10
#define FFH_AC_MODE_RELU     11
#define FFH_AC_MODE_SIGMOID  12
#define FFH_AC_MODE_TANH     13
#define FFH_AC_MODE_GELU     14
#define FFH_AGGR_MODE_NONE   20
#define FFH_AGGR_MODE_SUM    21
#define FFH_AGGR_MODE_AVG    22

/* limits */
#define FFH_MAX_TABLES        64   /* tables per batched embedding launch        */
#define FFH_MAX_CONCAT_INPUTS 256  /* [ref: include/config.h:30-37 MAX_NUM_INPUTS] */
#define FFH_MAX_SCRATCH_STREAMS 8  /* streams that may hold ctx-owned scratch at a time (ffh_ctx_reserve_scratch) */
/* block lengths of the canonical (two-level) summation order of the fused
 * embedding backward + SGD (see ffh_embedding_bwd_sgd_fused) */
#define FFH_EMB_CHUNK         32
#define FFH_EMB_CHUNK1        1024

typedef struct ffh_ctx ffh_ctx;   /* opaque; mirrors FFHandler [ref: include/config.h:75-84] */
typedef void* ffh_stream;         /* hipStream_t */
typedef void* ffh_event;          /* hipEvent_t  */
typedef void* ffh_graph;          /* hipGraphExec_t */

/* PerfMetrics subset kept on the device [ref: include/metrics_functions.h, src/metrics_functions/metrics_functions.cu:108-173] */
typedef struct ffh_perf_metrics {
  int32_t train_all;
  int32_t train_correct;
  float   cce_loss;
  float   sparse_cce_loss;
  float   mse_loss;
  float   rmse_loss;
  float   mae_loss;
  int32_t pad_;
} ffh_perf_metrics;

typedef struct ffh_device_info {
  char     name[128];
  char     arch[64];
  int32_t  compute_units;
  int32_t  wavefront_size;
  int64_t  total_mem_bytes;
  int32_t  lds_bytes_per_cu;
  int32_t  clock_khz;
} ffh_device_info;

/* One table of a batched (multi-table) embedding launch. */
typedef struct ffh_emb_table {
  const int64_t* idx;     /* [batch][in_dim] int64 row ids                            */
  float*         weight;  /* [num_entries][out_dim] fp32 table                        */
  float*         io;      /* fwd: out [batch][ld]; bwd: out_grad [batch][ld] (read)   */
  int64_t        num_entries;
  int64_t        ld;      /* leading dimension of `io` in floats (>= out_dim)         */
} ffh_emb_table;

/* ------------------------------------------------------------------ */
/* library / context                                                  */
/* ------------------------------------------------------------------ */
int         ffh_abi_version(void);
const char* ffh_backend_name(void);                 /* "hip-gfx950" | "oracle-cpu" */
int         ffh_ctx_create(ffh_ctx** out, int device);
int         ffh_ctx_destroy(ffh_ctx* ctx);
const char* ffh_last_error_string(const ffh_ctx* ctx);
int         ffh_device_query(ffh_ctx* ctx, ffh_device_info* info);
/* The library-owned ctx of the calling thread's CURRENT device (hipGetDevice), created on first use, one per device and
 * process, never destroyed by the caller.  For the reference's static kernels that receive NO handle or OpMeta:
 * Embedding::forward_kernel / backward_kernel [ref: include/model.h:1169-1186] and Concat::forward_kernel /
 * backward_kernel [ref: include/model.h:1771-1784] are static members without a meta argument (Linear and BatchMatmul carry
 * a `*Meta`, :1011-1027, :1098-1118), so their replacement bodies cannot reach FFHandler; they call this instead
 * (INTEGRATION.md section 2).  It has no workspace attached unless the caller attaches one. */
int         ffh_ctx_default(ffh_ctx** out);
/* attach caller-owned scratch; replaces FFHandler.workSpace/workSpaceSize */
int         ffh_ctx_set_workspace(ffh_ctx* ctx, void* ws, size_t bytes);
/* Math mode of the handle: cublasSetMathMode(handle.blas, CUBLAS_TENSOR_OP_MATH) behind --allow-tensor-op-math-conversion
 * [ref: src/runtime/model.cu:81-83; flag src/runtime/model.cc:2282-2403; call sites src/ops/linear.cu:436-453,624-659].
 *   FFH_MATH_DEFAULT          every GEMM in exact fp32 (v_mfma_f32_32x32x2_f32): the reference's default arithmetic
 *   FFH_MATH_TENSOR_OP_BF16   Linear GEMMs with in_dim >= FFH_BF16_MIN_DIM and out_dim >= FFH_BF16_MIN_DIM (forward, dX and
 *                             dW alike) round both operands to bfloat16 (nearest-even) as they are staged and multiply them
 *                             on v_mfma_f32_32x32x16_bf16: products are exact in fp32, accumulation is fp32, outputs, bias,
 *                             activations, master weights and the dW / db accumulators stay fp32.  Narrower layers, the
 *                             pair / skinny entry points, BatchMatmul and the dot interaction keep the fp32 kernels.
 * Tolerance of the bf16 mode against the fp32 result: |err| <= 2^-8 * sum_k |a_k b_k| per element (two roundings of 2^-9
 * relative each); against a twin that rounds the same operands (the oracle in the same mode) the fp32 summation-order
 * bound of the fp32 mode applies.  Returns FFH_ERR_BAD_ARG for an unknown mode. */
#define FFH_MATH_DEFAULT 0
#define FFH_MATH_TENSOR_OP_BF16 1
/*   FFH_MATH_FP32_SPLIT_BF16X3 the same layers, fp32-ACCURATE on the bf16 pipe (no reference counterpart; gfx950's fp32 MFMA
 *                             runs at 1/16 of the bf16 rate): each operand element is split into three bfloat16 terms
 *                             x1 + x2 + x3 (|x - x1 - x2 - x3| <= 2^-27 |x|) and a*b is the fp32 sum of the six exact
 *                             products a_i b_j with i + j <= 4 (dropped terms <= 2^-26 |a b|).  Held to the SAME tolerance
 *                             as FFH_MATH_DEFAULT (1e-5 of the term mass; measured error against float64 is of the size of
 *                             the exact-fp32 kernels' own).  Not bit-identical to the default mode; an INFINITE operand element
 *                             gives NaN outputs where fp32 arithmetic gives an infinity (x - x1 = inf - inf), and so does a
 *                             finite one above the largest bfloat16 (3.39e38); NaN stays NaN.  Opt-in. */
#define FFH_MATH_FP32_SPLIT_BF16X3 2
/*   FFH_MATH_FP32_SPLIT_BF16X3_ALL the split mode on EVERY Linear layer with both dims >= FFH_BF16_MIN_DIM, whatever its size (the layer rule of the mode as
 *                             round 5 shipped it): for tests and A/B runs -- small layers are slower here than on the fp32 kernels */
#define FFH_MATH_FP32_SPLIT_BF16X3_ALL 3
#define FFH_BF16_MIN_DIM 128
#define FFH_BF16X3_MIN_FLOP 1.0e10   /* FFH_MATH_FP32_SPLIT_BF16X3 takes the Linear layers with 2 * batch * in_dim * out_dim >= this (and both dims >= FFH_BF16_MIN_DIM): below, the exact-fp32 kernels are the faster fp32-accurate form */
int         ffh_ctx_set_math_mode(ffh_ctx* ctx, int mode);
/* bf16 MIRRORS for the tensor-op mode (no reference counterpart; the reference reaches its tensor cores through cuBLAS, which
 * converts internally).  In FFH_MATH_TENSOR_OP_BF16 the GEMM operands are rounded to bfloat16 anyway; with fp32 operands in HBM
 * the kernel is bound by bringing 4-byte elements on chip to use 2 of their bytes.  A caller may therefore give an fp32 buffer a
 * bfloat16 twin of the same element layout (same leading dimensions, 2 bytes per element):
 *   ffh_ctx_bf16_mirror_set(ctx, fp32_base, fp32_bytes, bf16_base)   registers [fp32_base, fp32_base + fp32_bytes) -> bf16_base
 *                                                                    (bf16_base NULL: removes the registration); <= 32 regions
 * From then on, in tensor-op mode only,
 *   producers that write fp32 values into a registered region also write their bf16 roundings (nearest even) into the twin:
 *     ffh_linear_fwd (y), ffh_linear_bwd / _ex (dx), when they run on the bf16 pipe (in_dim, out_dim >= FFH_BF16_MIN_DIM);
 *     ffh_embedding_fwd / _multi (out); ffh_sgd_update / _ex and ffh_adam_update (w); ffh_convert_f32_to_bf16 (explicit);
 *   the bf16-pipe GEMMs take an operand from its twin when BOTH operands of the GEMM lie in registered regions.
 * The twin of an element is by construction the value the kernel would have rounded it to, so results are bit-identical with and
 * without mirrors.  VALIDITY IS THE CALLER'S CONTRACT: register a region only if every writer of it is in the list above (or
 * refresh it with ffh_convert_f32_to_bf16); the library does not track who else writes the fp32 buffer.  A backward call that
 * applies a live activation derivative to dy in place (RELU / SIGMOID without FFH_LINEAR_DY_PREMASKED) does not read dy's twin. */
int         ffh_ctx_bf16_mirror_set(ffh_ctx* ctx, const void* fp32_base, size_t fp32_bytes, void* bf16_base);
int         ffh_convert_f32_to_bf16(ffh_ctx* ctx, void* dst_bf16, const float* src, int64_t count, ffh_stream s);
/* THREE-PLANE IMAGES for the fp32-accurate split mode (ABI 14; no reference counterpart).  In FFH_MATH_FP32_SPLIT_BF16X3 every GEMM operand
 * element is used as three bfloat16 terms x1 + x2 + x3; a kernel that splits fp32 operands as it stages them spends 3 vector instructions
 * per matrix instruction and moves every fp32 tile through registers (the split-in-kernel form, still the fall-back).  A caller may instead
 * give an fp32 buffer a PLANE IMAGE that its producers keep current, and the GEMMs then bring the terms on chip by LDS-DMA with no vector
 * work on the operand path (csrc/linear_x3_dma.hip):
 *   ffh_ctx_bf16x3_mirror_set(ctx, fp32_base, fp32_bytes, planes)   registers the region (planes NULL: removes it); fp32_base and planes
 *                                                                   128-byte aligned, planes holds FFH_BF16X3_IMAGE_BYTES(fp32_bytes)
 * Layout of the image ("I32"): 32 consecutive fp32 elements (128 bytes, counted from fp32_base) <-> 192 bytes
 *   [ x1 of the 32 | x2 of the 32 | x3 of the 32 ]   (64 bytes each; x1 = bf16(x) nearest-even, x2 = bf16(x - x1), x3 = bf16(x - x1 - x2),
 *                                                      both differences exact in fp32)
 * so element e (index from fp32_base) has its term p at byte (e / 32) * 192 + p * 64 + (e % 32) * 2.  A k-tile of 32 of a row is 192
 * contiguous bytes, 128 columns of a row 768: both operand orientations of the GEMMs stream whole runs.
 * From then on, in FFH_MATH_FP32_SPLIT_BF16X3 only,
 *   the producers listed for the bf16 twins above (ffh_linear_fwd: y; ffh_linear_bwd / _ex / _mse: dx; ffh_embedding_fwd / _multi: out, for
 *   leading dimensions that are multiples of 32; ffh_sgd_update / _ex, ffh_adam_update: w; ffh_convert_f32_to_bf16x3: explicit) keep the image of
 *   what they write into a registered region current -- the Linear entries under the same condition as for the twins: when the call runs on the
 *   mode's kernels (both dims >= FFH_BF16_MIN_DIM and, in FFH_MATH_FP32_SPLIT_BF16X3 proper, the FFH_BF16X3_MIN_FLOP rule; ffh_linear_last_route then
 *   names "bf16x3" or "x3_dma" for that GEMM).  A layer the mode leaves to the fp32 kernels writes fp32 only: refresh with ffh_convert_f32_to_bf16x3;
 *   a wide Linear GEMM takes BOTH operands from their images when both lie in registered regions, start a 32-element group and have leading
 *   dimensions that are multiples of 32 (reduction depth a multiple of 32); otherwise it splits in the kernel as before.
 * Same arithmetic either way (the six products per 32-deep k-step in the same order, fp32 accumulation): a forward / data-gradient result is
 * BIT-identical with and without images; weight gradients differ in the order of their split-K atomics only (and not even that
 * where the k-slices fit one round of workgroups on a stream with reserved scratch: they then meet through slots in slice order).  Validity is the caller's
 * contract, exactly as for the bf16 twins. */
#define FFH_BF16X3_IMAGE_BYTES(fp32_bytes) ((((size_t)(fp32_bytes) + 127) / 128) * 192)
int         ffh_ctx_bf16x3_mirror_set(ffh_ctx* ctx, const void* fp32_base, size_t fp32_bytes, void* planes);
/* recomputes the image of the rows x cols sub-matrix at src (leading dimension ld elements; rows == 1: a flat range) of a registered region;
 * any math mode.  FFH_ERR_BAD_ARG when [src, src + ((rows - 1) * ld + cols) * 4) is not inside a region registered with
 * ffh_ctx_bf16x3_mirror_set. */
int         ffh_convert_f32_to_bf16x3(ffh_ctx* ctx, const float* src, int64_t rows, int64_t cols, int64_t ld, ffh_stream s);
/* on != 0: every weight / bias gradient is produced WITHOUT floating-point atomics -- no split-K over workgroups (one
 * workgroup owns an output element and adds its k-ordered sum once), no per-workgroup partials meeting in one address;
 * the one-launch skinny / pair / dX+dW forms that rely on such atomics report FFH_ERR_UNSUPPORTED or are bypassed.  Results
 * are then bit-identical from run to run (the default mode differs in the last bits: fp32 atomic order), at a fraction of
 * the speed: a debugging mode for diffing two runs, the GPU counterpart of comparing against the sequential oracle.
 * (The embedding kernels are deterministic in both modes: their order is part of the ABI, FFH_EMB_CHUNK.) */
int         ffh_ctx_set_deterministic(ffh_ctx* ctx, int on);
/* Persistent weight-gradient GEMMs (one workgroup per CU for hundreds of microseconds) launched after this call leave `ncus` CUs
 * of the device without a workgroup (0: none, the default; rounded to a multiple of 8: one per XCD).  No reference counterpart -- a
 * scheduling hint of the kind Legion's mapper gives the reference: kernels that run BESIDE such a GEMM on other streams (the bottom
 * MLP's backward chain, the table update) are 5-10x slower on CUs they share with it than alone; a caller that knows such work is
 * pending trades a few percent of the GEMM for it.  Results do not depend on it beyond the stream-K partition (the usual fp32
 * summation-order bound).  Returns FFH_ERR_BAD_ARG for ncus < 0 or >= the device's CU count. */
int         ffh_ctx_set_dw_cu_reserve(ffh_ctx* ctx, int ncus);
/* Scratch the library owns for launches on stream `s` (ABI 12): the partial-tile slots + arrival counters of the stream-K forms with
 * fix-up (csrc/linear_sk.hip), the partial rows of the narrow-layer backward (csrc/linear.hip) and (round 6) 512 slots of 256 x 256 floats for the
 * k-slices of the bf16-pipe weight gradients (csrc/linear_x3_dma.hip, linear_bf16_dma.hip) -- ~170 MB.  THE one place they are
 * allocated: compute entry points never allocate (top of this file); a launch on a stream without scratch runs the other forms of the
 * same layers (same results to the usual bound; ffh_linear_last_route shows it).  Call it once per stream that runs Linear layers,
 * outside any capture; idempotent.  Released by ffh_stream_destroy(s) / ffh_ctx_destroy; at most FFH_MAX_SCRATCH_STREAMS streams hold
 * scratch at a time.  The reference's counterpart is the ones vector LinearMeta allocates per op and never frees
 * [ref: src/ops/linear.cu:986-994]. */
int         ffh_ctx_reserve_scratch(ffh_ctx* ctx, ffh_stream s);

/* memory / streams / events / graphs: what Legion+Realm provide to the
 * reference ops (regions, get_legion_stream [ref: src/runtime/cuda_helper.cu:5-31],
 * begin_trace/end_trace [ref: examples/cpp/DLRM/dlrm.cc:174-181]). */
int ffh_malloc(ffh_ctx* ctx, void** ptr, size_t bytes);
int ffh_free(ffh_ctx* ctx, void* ptr);
int ffh_memcpy_h2d(ffh_ctx* ctx, void* dst, const void* src, size_t bytes, ffh_stream s);
int ffh_memcpy_d2h(ffh_ctx* ctx, void* dst, const void* src, size_t bytes, ffh_stream s);
int ffh_memcpy_d2d(ffh_ctx* ctx, void* dst, const void* src, size_t bytes, ffh_stream s);
int ffh_stream_create(ffh_ctx* ctx, ffh_stream* s);
/* ABI 11: the same with a HIP stream priority (0 = default, negative = higher, positive = lower; clamped to the device's range) --
 * what the Legion mapper's task priorities are to the reference.  Optional; the DLRM shim uses it behind --stream-priorities only
 * (measured level on the single-GPU step, and harmful on a stream that carries RCCL's send/recv kernels). */
int ffh_stream_create_with_priority(ffh_ctx* ctx, ffh_stream* s, int priority);
int ffh_stream_destroy(ffh_ctx* ctx, ffh_stream s);
int ffh_stream_sync(ffh_ctx* ctx, ffh_stream s);
int ffh_device_sync(ffh_ctx* ctx);
int ffh_event_create(ffh_ctx* ctx, ffh_event* e);
/* an event that only orders streams: no timestamps (ffh_event_elapsed_ms refuses it), cheaper to record */
int ffh_event_create_sync(ffh_ctx* ctx, ffh_event* e);
int ffh_event_destroy(ffh_ctx* ctx, ffh_event e);
int ffh_event_record(ffh_ctx* ctx, ffh_event e, ffh_stream s);
int ffh_event_sync(ffh_ctx* ctx, ffh_event e);
int ffh_stream_wait_event(ffh_ctx* ctx, ffh_stream s, ffh_event e);
int ffh_event_elapsed_ms(ffh_ctx* ctx, ffh_event start, ffh_event stop, float* ms);
int ffh_graph_begin_capture(ffh_ctx* ctx, ffh_stream s);
int ffh_graph_end_capture(ffh_ctx* ctx, ffh_stream s, ffh_graph* g);
int ffh_graph_launch(ffh_ctx* ctx, ffh_graph g, ffh_stream s);
int ffh_graph_destroy(ffh_ctx* ctx, ffh_graph g);

/* ------------------------------------------------------------------ */
/* initialisers and synthetic data (counter-based, seeded, bit-exact    */
/* between the two backends; see DESIGN.md "RNG")                       */
/* ------------------------------------------------------------------ */
/* assign_kernel [ref: src/runtime/cuda_helper.cu:52-60]; ZeroInitializer
 * [ref: src/runtime/initializer_kernel.cu:209-241] is value = 0 */
int ffh_fill_f32(ffh_ctx* ctx, float* ptr, int64_t count, float value, ffh_stream s);
int ffh_zero(ffh_ctx* ctx, void* ptr, size_t bytes, ffh_stream s);
/* UniformInitializer [ref: src/runtime/initializer_kernel.cu:24-98] (own stream, not cuRAND's):
 * ptr[i] = lo + (hi-lo) * u24(seed, i) */
int ffh_init_uniform(ffh_ctx* ctx, float* ptr, int64_t count, uint64_t seed, float lo, float hi, ffh_stream s);
/* synthetic DLRM inputs, distributions of [ref: examples/cpp/DLRM/dlrm.cc:413-420]:
 * idx[i] = hash(seed, first + i) mod num_entries; dense = u24 in [0,1); label in {0,1} */
int ffh_gen_indices(ffh_ctx* ctx, int64_t* idx, int64_t count, uint64_t seed, int64_t first, int64_t num_entries, ffh_stream s);
int ffh_gen_uniform01(ffh_ctx* ctx, float* ptr, int64_t count, uint64_t seed, int64_t first, ffh_stream s);
int ffh_gen_bernoulli(ffh_ctx* ctx, float* ptr, int64_t count, uint64_t seed, int64_t first, ffh_stream s);

/* ------------------------------------------------------------------ */
/* Embedding                                                          */
/* ------------------------------------------------------------------ */
/* Embedding::forward_kernel [ref: include/model.h:1169-1177, src/ops/embedding.cu:219-231,166-190]
 * out[b][d] = sum_{j<in_dim} weight[idx[b][j]][d], j ascending, fp32, starting from +0.
 * aggr: FFH_AGGR_MODE_SUM, or FFH_AGGR_MODE_AVG (true mean: sum * (1/in_dim); the
 * reference's AVG divides inside the j loop, a bug not reproduced -- SURVEY 8a-1).
 * Row ids must satisfy 0 <= idx < num_entries (CPU reference asserts it,
 * [ref: src/ops/embedding.cc:71-73]); out-of-range ids are not checked on the device.
 * The reference's static signature carries no row count (its GPU kernel never reads one, [ref: src/ops/embedding.cu:166-190]):
 * a caller that does not know it passes FFH_NUM_ENTRIES_UNKNOWN to ffh_embedding_fwd / ffh_embedding_bwd_dense, which use
 * num_entries for argument validation only (the fused update does need the real count: it sizes the radix sort). */
#define FFH_NUM_ENTRIES_UNKNOWN ((int64_t)1 << 40)
int ffh_embedding_fwd(ffh_ctx* ctx, const int64_t* idx, float* out, const float* weight,
                      int in_dim, int out_dim, int64_t batch, int64_t num_entries,
                      int64_t out_ld, int aggr, ffh_stream s);
/* the same for n tables in one launch (tables[i].io = out) */
int ffh_embedding_fwd_multi(ffh_ctx* ctx, const ffh_emb_table* tables, int ntables,
                            int in_dim, int out_dim, int64_t batch, int aggr, ffh_stream s);

/* Embedding::backward_kernel [ref: include/model.h:1178-1186, src/ops/embedding.cu:308-320,192-217]
 * weight_grad[idx[b][j]][d] += out_grad[b][d] (AVG: / in_dim) with fp32 atomics into the
 * dense, pre-zeroed full-table gradient -- the reference's own backward. */
int ffh_embedding_bwd_dense(ffh_ctx* ctx, const int64_t* idx, const float* out_grad, float* weight_grad,
                            int in_dim, int out_dim, int64_t batch, int64_t num_entries,
                            int64_t grad_ld, int aggr, ffh_stream s);

/* Fused embedding backward + SGD: the net effect of
 *   Op::zero_grad            [ref: src/runtime/model.cc:466-490]
 *   Embedding::backward_kernel [ref: src/ops/embedding.cu:192-217]
 *   sgd_update (momentum 0, weight_decay 0) [ref: src/runtime/optimizer_kernel.cu:23-41]
 * on one table, without the dense gradient:
 *   weight[r][:] -= lr * sum_{(b,j): idx[b][j]==r} out_grad[b][:]      (rows not hit: untouched)
 * Canonical (deterministic) summation order, chosen so that a hot row is summed by many lane-groups
 * at once yet the result never depends on the launch geometry: the table's contributions are sorted
 * by (row, position p = b*in_dim+j); the sorted list is cut at multiples of FFH_EMB_CHUNK (32) and of
 * FFH_EMB_CHUNK1 (1024).  Inside a 32-block the row's contributions are added left to right in fp32;
 * the 32-block partials of a row inside one 1024-block are added left to right; the 1024-block
 * partials of the row are added left to right; then one `w = fmaf(-lr, sum, w)`.
 * Needs ffh_embedding_bwd_workspace_bytes() of workspace attached to the ctx. */
int ffh_embedding_bwd_sgd_fused(ffh_ctx* ctx, const int64_t* idx, const float* out_grad, float* weight,
                                int in_dim, int out_dim, int64_t batch, int64_t num_entries,
                                int64_t grad_ld, int aggr, float lr, ffh_stream s);
int ffh_embedding_bwd_sgd_fused_multi(ffh_ctx* ctx, const ffh_emb_table* tables, int ntables,
                                      int in_dim, int out_dim, int64_t batch, int aggr, float lr, ffh_stream s);
size_t ffh_embedding_bwd_workspace_bytes(int ntables, int in_dim, int out_dim, int64_t batch);
/* The same update in two calls (ABI 7): the stable sort by (row, position) reads only the indices, so a caller that has them
 * before the output gradients exist (a training step: from the gather on) can run it early --
 *   ffh_embedding_bwd_sort_multi      the index-only part, into the ctx workspace (tables[].io / .ld are not read);
 *   ffh_embedding_bwd_sgd_apply_multi the rest (segmented sums, folds, w = fmaf(-lr, sum, w)), on the SAME tables / indices /
 *                                     batch, with nothing else using the workspace in between and ordered behind the sort by
 *                                     the caller's streams / events.
 * sort + apply launch what ffh_embedding_bwd_sgd_fused_multi launches, in the same order: the same bits.
 * ONE-SHOT: the apply phase consumes the sort (the sorted list AND the fold counters / level-1 slots that only the sort clears).
 * A ctx remembers (workspace, ntables, in_dim, out_dim, batch) of its latest sort; apply without such a note -- no sort, a second
 * apply, another shape, another workspace attached, or a fused call on that workspace in between -- returns FFH_ERR_WORKSPACE and
 * launches nothing.  (Both calls on the same ctx; a capture that records sort and apply replays them as a pair.) */
int ffh_embedding_bwd_sort_multi(ffh_ctx* ctx, const ffh_emb_table* tables, int ntables,
                                 int in_dim, int out_dim, int64_t batch, ffh_stream s);
int ffh_embedding_bwd_sgd_apply_multi(ffh_ctx* ctx, const ffh_emb_table* tables, int ntables,
                                      int in_dim, int out_dim, int64_t batch, int aggr, float lr, ffh_stream s);
/* Which form the most recent fused table update on this ctx took (diagnostic, like ffh_linear_last_route; the result is the same
 * bits in every form): "small" (one launch, <= 2048 lookups per table), "lsd:passes=P" (P stable LSD passes + the apply launch), or
 * "buckets:bits=M" (one stable pass on the top M id bits, the rest of the order made per tile inside the apply launch: three launches;
 * calls of <= 64 K lookups per table whose ids need more than one pass).  The oracle returns "oracle". */
const char* ffh_embedding_last_route(const ffh_ctx* ctx);

/* Any optimizer on the sorted segments (ABI 8; SURVEY 8f-4 "needs per-row state for the sparse variant").  The reference runs
 * sgd_update / adam_update over EVERY element of a parameter [ref: src/runtime/optimizer_kernel.cu:23-41,206-226] -- for an embedding
 * table a dense gradient plus three or five full-table sweeps per step.  These entry points apply the same element arithmetic
 * (statement by statement that of ffh_sgd_update_ex / ffh_adam_update) to the rows a batch touched, with the row's gradient = its
 * canonical sum (FFH_EMB_CHUNK order, as the fused update), and leave every other row AND its optimizer state untouched:
 *   FFH_SPARSE_OPT_SGD           w = fmaf(-lr, g, w)                                   == ffh_embedding_bwd_sgd_fused_multi
 *   FFH_SPARSE_OPT_SGD_MOMENTUM  gt = g + wd w; V = V mom + gt; gt = nesterov ? gt + mom V : V (mom > 0); w -= lr gt     state s0 = V
 *   FFH_SPARSE_OPT_ADAM          gt = g + wd w; M = b1 M + (1-b1) gt; V = b2 V + (1-b2) gt gt; w -= lr M / (sqrt(V) + eps)
 *                                (lr = alpha_t, advanced by the caller as for ffh_adam_update)                state s0 = M, s1 = V
 * STATED DIVERGENCE from the reference's dense sweep ("lazy" semantics, those of torch.optim.SparseAdam / sparse SGD): a row the
 * batch does not touch keeps w, M, V -- the dense sweep would decay its moments (and move w by the decayed momentum, and by
 * wd w when weight_decay != 0) every step.  Equal to the dense sweep on every row of a step in which the state of the untouched
 * rows is zero and weight_decay == 0 (e.g. the first step); on touched rows always, given the same row gradient.  A caller that
 * needs the reference's semantics on every row uses ffh_embedding_bwd_dense + ffh_sgd_update / ffh_adam_update on a dense gradient.
 * states[i].s0 / .s1: [num_entries][out_dim] fp32 like tables[i].weight (null where the kind has no such state).
 * _fused_multi sorts and applies; _apply_multi is the second half of the two-call form behind ffh_embedding_bwd_sort_multi. */
#define FFH_SPARSE_OPT_SGD          0
#define FFH_SPARSE_OPT_SGD_MOMENTUM 1
#define FFH_SPARSE_OPT_ADAM         2
typedef struct ffh_sparse_opt {
  int32_t kind;            /* FFH_SPARSE_OPT_*                          */
  float   lr;              /* SGD: learning rate; Adam: alpha_t          */
  float   weight_decay;
  float   momentum;        /* FFH_SPARSE_OPT_SGD_MOMENTUM               */
  int32_t nesterov;
  float   beta1, beta2, epsilon;   /* FFH_SPARSE_OPT_ADAM               */
} ffh_sparse_opt;
typedef struct ffh_emb_state { float* s0; float* s1; } ffh_emb_state;
int ffh_embedding_bwd_opt_fused_multi(ffh_ctx* ctx, const ffh_emb_table* tables, const ffh_emb_state* states, int ntables,
                                      int in_dim, int out_dim, int64_t batch, int aggr, const ffh_sparse_opt* opt, ffh_stream s);
int ffh_embedding_bwd_opt_apply_multi(ffh_ctx* ctx, const ffh_emb_table* tables, const ffh_emb_state* states, int ntables,
                                      int in_dim, int out_dim, int64_t batch, int aggr, const ffh_sparse_opt* opt, ffh_stream s);

/* Row-wise sharded table (no reference counterpart: the reference splits an embedding on the sample dim only,
 * [ref: src/ops/embedding.cu:84-85]).  A rank holds rows [row_begin, row_begin + rows_local) followed by ONE extra
 * all-zero row; this maps global ids to local ones: local[i] = idx[i] - row_begin when the row is held here, else
 * rows_local (the zero row: the gather then adds +0 for it, and the update's writes to it are discarded by the caller
 * clearing the row again).  idx and local may be the same buffer. */
int ffh_embedding_localize_rows(ffh_ctx* ctx, const int64_t* idx, int64_t* local, int64_t count,
                                int64_t row_begin, int64_t rows_local, ffh_stream s);

/* ------------------------------------------------------------------ */
/* Linear                                                             */
/* ------------------------------------------------------------------ */
/* Linear::forward_kernel [ref: include/model.h:1011-1017, src/ops/linear.cu:425-465]
 * y[b][o] = act( sum_i x[b][i]*w[o][i] + bias[o] ); bias may be NULL; fp32 throughout
 * (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32). activation: NONE, RELU, SIGMOID, and GELU forward-only as in the reference
 * (tanh form, [ref: src/ops/linear.cu:454-459; its backward asserts NONE / RELU / SIGMOID, :632-635]); TANH: unsupported. */
int ffh_linear_fwd(ffh_ctx* ctx, const float* x, int64_t ldx, float* y, int64_t ldy,
                   const float* w, const float* bias,
                   int in_dim, int out_dim, int64_t batch, int activation, ffh_stream s);
/* FAST-PATH CONTRACT of ffh_linear_fwd / ffh_linear_bwd* (every shape is served; this is what the persistent MFMA kernels of
 * csrc/linear_sk.hip take -- 140-150 TFLOP/s against 75-100 on the general kernels): in_dim % 64 == 0, out_dim % 128 == 0, batch %
 * 128 == 0, every operand row 16-byte aligned (pointers and leading dimensions multiples of 4 floats).  A layer whose in_dim breaks
 * the first rule (MLPerf-DLRM's 479-wide first top layer: rows start at odd dwords, K is not whole k-tiles) reaches it by ZERO
 * PADDING THE REDUCTION DEPTH in its caller's allocation: x [batch][ld = P] and w [out][ld = P] with P = ffh_linear_fast_in_dim(in_dim,
 * out_dim), columns in_dim .. P - 1 zero, and in_dim = P in the calls.  Same values: the pads add exact zeros at the end of every k
 * sum; dw's pad columns are dy^T * 0 = 0, so zero-initialised pads stay zero under SGD / momentum / weight decay / Adam; dx's pad columns
 * are dy * 0 and nobody reads them.  The bundled shim does exactly this (FFModel::allocate step 4a; 8192 x 479 -> 1024: 99.6 / 74.9 / 75.2
 * TFLOP/s unpadded, 123 / 130 / 111 padded).  Returns in_dim itself where padding does not pay (narrow layers, already a multiple). */
int ffh_linear_fast_in_dim(int in_dim, int out_dim);
/* Which kernel families the most recent ffh_linear_* call on this ctx launched: ';'-separated tokens "<gemm>:<family>[:detail]"
 * with <gemm> in {fwd, dx, dw, bwd} (e.g. "dx:f32_128x128x16;dw:glds_64x64:splitk=8").  Diagnostic (no reference
 * counterpart): lets a test assert that a shape really took the route it is meant to cover.  The oracle returns "oracle". */
const char* ffh_linear_last_route(const ffh_ctx* ctx);
/* Linear::backward_kernel [ref: include/model.h:1018-1027, src/ops/linear.cu:610-660]
 * in place: dy = dy * act'(y)   (relu: y>0 ? dy : 0 [ref: src/runtime/cuda_helper.cu:71-78];
 *                                sigmoid: dy*y*(1-y) [ref: src/ops/linear.cu:600-607])
 * dw[o][i] += sum_b dy[b][o]*x[b][i];  db[o] += sum_b dy[b][o] (db may be NULL);
 * dx[b][i] += sum_o dy[b][o]*w[o][i]   (dx may be NULL: first layer, gradient discarded).
 * All three accumulate (beta = 1) into buffers the caller zeroed, exactly as the reference. */
int ffh_linear_bwd(ffh_ctx* ctx, const float* x, int64_t ldx, float* dx, int64_t lddx,
                   const float* y, int64_t ldy, float* dy, int64_t lddy,
                   const float* w, float* dw, float* db,
                   int in_dim, int out_dim, int64_t batch, int activation, ffh_stream s);

/* Linear::backward_kernel with the scheduling freedoms the Legion task graph gives the reference
 * (independent tasks run concurrently) made explicit.  Same arithmetic as ffh_linear_bwd; flags:
 *   FFH_LINEAR_DX_OVERWRITE  dx = dy*w instead of dx += dy*w (caller knows dx has no other producer,
 *                            so it need not be zeroed first: 0 + x == x)
 *   s_dw != NULL, != s       the weight/bias-gradient GEMM is issued on stream s_dw behind an event
 *                            recorded on s; the data-gradient GEMM on s does not wait for it (it applies
 *                            relu' to dy as it loads it instead of reading the in-place result).  The
 *                            caller joins s_dw before it consumes dw/db.  dy still ends up overwritten
 *                            by the activation gradient, as in the reference. */
#define FFH_LINEAR_DX_OVERWRITE 1
/* split form, for callers that issue the two GEMMs themselves (e.g. from two host threads on two streams):
 *   FFH_LINEAR_ONLY_DX  [sigmoid: in-place activation gradient + db first] then dx only; relu' is applied to dy
 *                       while it is loaded, dy itself is not written
 *   FFH_LINEAR_ONLY_DW  dw (+ db for relu / none) only; relu' is applied on load and written back to dy in place.
 *                       For sigmoid the ONLY_DX call must have completed on the device first.
 * Issued once each (any order for relu / none) they equal one ffh_linear_bwd call. */
#define FFH_LINEAR_ONLY_DX 4
#define FFH_LINEAR_ONLY_DW 2
/* moving relu' to the producer of a gradient (what a fused Linear->Linear chain does): the layer ABOVE hands down a
 * gradient that already carries this layer's activation derivative, so no kernel has to read y and dy again:
 *   FFH_LINEAR_DX_MASK_BY_X  x is the output of a ReLU (the layer below): the data gradient this call produces is
 *                            (x > 0) ? dy*w : 0, i.e. reluBackward [ref: src/runtime/cuda_helper.cu:71-78] of the layer
 *                            below applied where its operand is produced (x > 0 <=> that layer's y > 0)
 *   FFH_LINEAR_DY_PREMASKED  dy already carries this layer's activation derivative (its consumer ran with
 *                            DX_MASK_BY_X): that step is skipped, dy is not modified, db = column sums of dy
 * A layer run with DY_PREMASKED below a layer run with DX_MASK_BY_X computes exactly what two plain calls compute. */
#define FFH_LINEAR_DY_PREMASKED 8
#define FFH_LINEAR_DX_MASK_BY_X 16
/* ffh_linear_bwd_ex is free NOT to use s_dw (e.g. when it issues both GEMMs of a mid-size layer as one launch on s).
 * Returns 1 if any call on this ctx has issued work on a caller-supplied second stream since the flag was last
 * cleared (clear != 0 clears it), else 0: a caller that joins s_dw only when this says so saves the join's packets. */
int ffh_second_stream_used(ffh_ctx* ctx, int clear);
/* The next ffh_linear_bwd_ex on this ctx also records `e` on its stream s, behind everything it puts there -- the same
 * as calling ffh_event_record(ctx, e, s) right after it, except that the library may hang the event on its last kernel's
 * own completion signal (hipExtLaunchKernelGGL's stop event) instead of sending a separate barrier packet down s. */
int ffh_event_record_with_next_linear_bwd(ffh_ctx* ctx, ffh_event e);
/* Storing a data gradient where a Concat backward would copy it afterwards.  The NEXT ffh_linear_bwd_ex on this ctx, if it is
 * called with FFH_LINEAR_DX_OVERWRITE for in_dim == ncols and runs as the one-launch LDS-DMA form or as the register-staged
 * fp32 data-gradient GEMM (not: skinny / pair launches, the bf16-pipe math modes, deterministic mode), writes column n of dX to
 * map[n].base[row * map[n].ld] instead of dx[row * lddx + n] (map: device memory, ncols entries, must stay valid until that
 * call's kernels have run).  `attach_if_used` (may be NULL) is then signalled behind that launch, like
 * ffh_event_record_with_next_linear_bwd; if the call cannot take the map it writes dx as always and records nothing.
 * ffh_linear_dx_scatter_used() says which of the two happened (it reports on the last such call). */
typedef struct ffh_col_dest { float* base; int64_t ld; } ffh_col_dest;
int ffh_linear_bwd_set_dx_scatter(ffh_ctx* ctx, const ffh_col_dest* map, int ncols, ffh_event attach_if_used);
int ffh_linear_dx_scatter_used(ffh_ctx* ctx);
/* The bias gradient of the layer BELOW as a by-product of this layer's data gradient (ABI 10).  In a Linear -> Linear chain the dx a
 * layer stores (FFH_LINEAR_DX_OVERWRITE; with FFH_LINEAR_DX_MASK_BY_X when the layer below ends in a ReLU) IS the lower layer's final dy,
 * and that layer's db is the column sums of it [ref: the cublasSgemv over dy, src/ops/linear.cu:644-651].  Taken inside the lower
 * layer's weight-gradient GEMM those sums cost it 6 % (every k-tile of dy is summed again in every tile column); taken where the tile
 * of dx is complete and in registers -- the epilogue of this layer's data-gradient kernel -- they are 64 adds per tile.
 * The NEXT ffh_linear_bwd / _ex on this ctx, if it stores its data gradient (DX_OVERWRITE, in_dim == ncols) through the persistent
 * fp32 kernel with the plain store epilogue (not: column map pending, deterministic mode, other kernels), also does
 * colsum[n] += sum over rows of dx[row][n] (after the mask).  One call only, taken or not; ffh_linear_dx_colsum_used() says whether the
 * last such call took it -- then the caller passes db = NULL to the lower layer's call, else it passes db as always. */
int ffh_linear_bwd_set_dx_colsum(ffh_ctx* ctx, float* colsum, int ncols);
int ffh_linear_dx_colsum_used(ffh_ctx* ctx);
int ffh_linear_bwd_ex(ffh_ctx* ctx, const float* x, int64_t ldx, float* dx, int64_t lddx,
                      const float* y, int64_t ldy, float* dy, int64_t lddy,
                      const float* w, float* dw, float* db,
                      int in_dim, int out_dim, int64_t batch, int activation,
                      int flags, ffh_stream s, ffh_stream s_dw);

/* ------------------------------------------------------------------ */
/* Concat                                                             */
/* ------------------------------------------------------------------ */
/* Concat::forward_kernel [ref: include/model.h:1771-1777, src/ops/concat.cu:211-249] with the
 * Domain arguments flattened the way calc_blk_size does [ref: src/ops/concat.cu:194-208]:
 * for every block blk < num_blocks: out[blk*out_blk + off_i + e] = in_i[blk*in_ld[i] + e],
 * e < in_blk[i], off_i = sum_{i'<i} in_blk[i'].  in_ld may be NULL (= in_blk).  One launch
 * for all inputs (the reference launches copy_with_stride once per input,
 * [ref: src/runtime/cuda_helper.cu:128-144]).  An input whose pointer already is
 * out + off_i with in_ld[i] == out_blk is skipped (aliased producer). */
int ffh_concat_fwd(ffh_ctx* ctx, float* out, int64_t out_blk, const float* const* ins,
                   const int64_t* in_blk, const int64_t* in_ld, int num_inputs,
                   int64_t num_blocks, ffh_stream s);
/* Concat::backward_kernel [ref: include/model.h:1778-1784, src/ops/concat.cu:325-360],
 * add_with_stride [ref: src/runtime/cuda_helper.cu:110-126]:
 * in_grad_i[blk*in_ld[i] + e] += out_grad[blk*out_blk + off_i + e]  (accumulate) */
int ffh_concat_bwd(ffh_ctx* ctx, const float* out_grad, int64_t out_blk, float* const* in_grads,
                   const int64_t* in_blk, const int64_t* in_ld, int num_inputs,
                   int64_t num_blocks, ffh_stream s);

/* The same with flags.  FFH_CONCAT_BWD_OVERWRITE: in_grad_i = slice (stored, not accumulated) -- for inputs whose gradient
 * has no other producer, so that it need not be zeroed first (0 + x == x), as FFH_LINEAR_DX_OVERWRITE. */
#define FFH_CONCAT_BWD_OVERWRITE 1
int ffh_concat_bwd_ex(ffh_ctx* ctx, const float* out_grad, int64_t out_blk, float* const* in_grads,
                      const int64_t* in_blk, const int64_t* in_ld, int num_inputs,
                      int64_t num_blocks, int flags, ffh_stream s);

/* ------------------------------------------------------------------ */
/* BatchMatmul                                                        */
/* ------------------------------------------------------------------ */
/* BatchMatmul::forward_kernel [ref: include/model.h:1098-1108, src/ops/batch_matmul.cu:194-244]
 * A [batch][n][k], B [batch][k][m], O [batch][n][m]; O = A*B.  a_seq_length_dim /
 * b_seq_length_dim / seq_length shrink k, n or m as the reference does (-1: unused). */
int ffh_bmm_fwd(ffh_ctx* ctx, float* o, const float* a, const float* b,
                int m, int n, int k, int64_t batch,
                int a_seq_length_dim, int b_seq_length_dim, int seq_length, ffh_stream s);
/* BatchMatmul::backward_kernel [ref: include/model.h:1109-1118, src/ops/batch_matmul.cu:375-400]
 * a_grad += o_grad * B^T ; b_grad += A^T * o_grad  (both accumulate) */
int ffh_bmm_bwd(ffh_ctx* ctx, const float* o_grad, const float* a, float* a_grad,
                const float* b, float* b_grad, int m, int n, int k, int64_t batch, ffh_stream s);

/* ------------------------------------------------------------------ */
/* Transpose (the reference-op composition of the dot interaction) */
/* ------------------------------------------------------------------ */
/* Transpose::forward_kernel [ref: src/ops/transpose.cu:195-251]: out = permute(in): out.dims[i] = in.dims[perm[i]]
 * (dims and perm in natural order, batch first; ndim <= 4).  The reference kernel adds into the output
 * (`out += out*beta + in` with beta = 0, a slip); the restated semantics is the plain permutation. */
int ffh_transpose_fwd(ffh_ctx* ctx, float* out, const float* in, int ndim, const int64_t* in_dims, const int* perm, ffh_stream s);
/* Transpose::backward_kernel [ref: src/ops/transpose.cu:262-330]: in_grad += inverse-permute(out_grad) */
int ffh_transpose_bwd(ffh_ctx* ctx, float* in_grad, const float* out_grad, int ndim, const int64_t* in_dims, const int* perm, ffh_stream s);

/* The last layer's backward with the MSE loss step folded in: exactly
 *   ffh_mse_bwd_metrics(ctx, dy, y, label, perf, batch, out_dim, scale, metrics_flags, s)   (dy = (y - label) * scale, metrics)
 *   ffh_linear_bwd_ex(ctx, x, ..., activation, flags, s, s_dw)
 * as ONE launch (the loss gradient never goes through memory on its own; dy ends up holding what the two calls leave
 * there).  Only for the layers the one-launch backward serves (out_dim <= 4 and in_dim <= 1024, 16-byte aligned
 * operands) without FFH_LINEAR_ONLY_* / FFH_LINEAR_DY_PREMASKED: anything else returns FFH_ERR_UNSUPPORTED and nothing has
 * been launched -- the caller then makes the two calls.  label is [batch][out_dim] contiguous. */
int ffh_linear_bwd_mse(ffh_ctx* ctx, const float* x, int64_t ldx, float* dx, int64_t lddx,
                       const float* y, int64_t ldy, float* dy, int64_t lddy,
                       const float* w, float* dw, float* db,
                       int in_dim, int out_dim, int64_t batch, int activation, int flags,
                       const float* label, float scale, ffh_perf_metrics* perf, int metrics_flags, ffh_stream s);

/* Two narrow layers at the end of a chain (DLRM's bottom MLP ends 256 -> 64 -> 16), x_l -> [LOWER: W_l, act_l] -> x_u ->
 * [UPPER: W_u, act_u] -> y_u: the upper layer's whole backward and the lower layer's data gradient as ONE launch.  Exactly
 *   ffh_linear_bwd_ex(ctx, x_u, ldx_u, dy_l, lddy_l, y_u, ldy_u, dy_u, lddy_u, w_u, dw_u, db_u, in_u, out_u, batch, act_u,
 *                     flags_u | FFH_LINEAR_DX_OVERWRITE | (act_l == RELU ? FFH_LINEAR_DX_MASK_BY_X : 0), s, NULL)
 *   ffh_linear_bwd_ex(ctx, x_l, ldx_l, dx_l, lddx_l, x_u, ldx_u, dy_l, lddy_l, w_l, NULL, NULL, in_l, in_u, batch, act_l,
 *                     flags_l | FFH_LINEAR_ONLY_DX | FFH_LINEAR_DY_PREMASKED, s, NULL)
 * (dy_l, the gradient between the layers, is written with the lower layer's activation derivative applied; the lower
 * layer's dW / db remain one ffh_linear_bwd_ex(..., FFH_LINEAR_ONLY_DW | FFH_LINEAR_DY_PREMASKED) over the batch).
 * flags_u: FFH_LINEAR_DY_PREMASKED or 0; flags_l: FFH_LINEAR_DX_OVERWRITE, FFH_LINEAR_DX_MASK_BY_X.  Served shapes:
 * out_u <= 16, in_u 32 or 64, in_l a multiple of 32, act_l RELU or NONE; anything else returns FFH_ERR_UNSUPPORTED with
 * nothing launched and the caller makes the two calls.  w_u is [out_u][in_u], w_l is [in_u][in_l]. */
int ffh_linear_pair_bwd(ffh_ctx* ctx, const float* x_u, int64_t ldx_u, const float* y_u, int64_t ldy_u, float* dy_u, int64_t lddy_u,
                        const float* w_u, float* dw_u, float* db_u, int in_u, int out_u, int act_u, int flags_u,
                        const float* x_l, int64_t ldx_l, float* dx_l, int64_t lddx_l, float* dy_l, int64_t lddy_l, const float* w_l,
                        int in_l, int act_l, int flags_l, int64_t batch, ffh_stream s);
/* ... and the forward of the same two layers as one launch: exactly
 *   ffh_linear_fwd(ctx, x_l, ldx_l, y_l, ldy_l, w_l, b_l, in_l, mid, batch, act_l, s)
 *   ffh_linear_fwd(ctx, y_l, ldy_l, y_u, ldy_u, w_u, b_u, mid, out_u, batch, act_u, s)
 * for mid 32 or 64, out_u <= 16, in_l a multiple of 128 (mid 64) / 256 (mid 32), 16-byte aligned x_l / w_l / w_u rows;
 * anything else: FFH_ERR_UNSUPPORTED, nothing launched.  w_l is [mid][in_l], w_u is [out_u][mid]. */
int ffh_linear_pair_fwd(ffh_ctx* ctx, const float* x_l, int64_t ldx_l, const float* w_l, const float* b_l, int in_l, int act_l,
                        float* y_l, int64_t ldy_l, int mid, const float* w_u, const float* b_u, int out_u, int act_u,
                        float* y_u, int64_t ldy_u, int64_t batch, ffh_stream s);

/* A CHAIN of narrow Linear layers, x -> [L0] -> y0 -> [L1] -> ... -> y(n-1), every width <= FFH_CHAIN_MAX_WIDTH (DLRM's bottom MLP
 * 13-512-256-128; the Kaggle shape's 13-512-256-64-16 and 432-512-256-1): what nlayers calls of Linear::forward_kernel /
 * backward_kernel compute [ref: src/ops/linear.cu:425-465,610-660], as one launch forward and two launches backward -- the
 * activations between the layers stay in LDS, the weights stream from L2 into MFMA operand registers (csrc/mlp_chain.hip).  The
 * reference's precedent for several operators in one task is FusedOp [ref: src/ops/fused.cu:283-400].  Layer l: w [out][ldw],
 * bias [out] or NULL, y [batch][ldy] its output, dy [batch][lddy] the gradient of the loss with respect to y, dw [out][ldw],
 * db [out] or NULL (backward only: dy, dw, db may be NULL forward).
 *   ffh_mlp_chain_fwd  ==  for l = 0 .. n-1: ffh_linear_fwd(x_l, ldx_l, y_l, ldy_l, w_l, bias_l, in_l, out_l, batch, act_l)
 *                          with x_0 = x, x_l = y_(l-1); every y_l is written (the backward reads it).
 *   ffh_mlp_chain_bwd  ==  for l = n-1 .. 0: ffh_linear_bwd_ex(x_l, ldx_l, dx_l, .., y_l, ldy_l, dy_l, lddy_l, w_l, dw_l, db_l, in_l, out_l,
 *                          batch, act_l, flags_l, s, NULL) with dx_l = dy_(l-1) stored (FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X
 *                          where layer l-1 ends in a ReLU, and then FFH_LINEAR_DY_PREMASKED for layer l-1), dx_0 = dx (may be NULL: the
 *                          gradient is discarded).  `flags`: FFH_LINEAR_DY_PREMASKED applies to the TOP layer's dy (else its activation
 *                          derivative is applied to dy in place, as the reference leaves it), FFH_LINEAR_DX_OVERWRITE and
 *                          FFH_LINEAR_DX_MASK_BY_X to dx.  dw / db accumulate into buffers the caller zeroed (atomics: not in
 *                          deterministic mode).  An event attached with ffh_event_record_with_next_linear_bwd is recorded behind the
 *                          data-gradient chain, in front of the weight gradients (dx and every dy are final there).
 * Served: fp32 math mode -- and FFH_MATH_FP32_SPLIT_BF16X3 (proper, not _ALL) when every layer of the chain is one that mode leaves to the exact
 * kernels (2 * batch * in * out < FFH_BF16X3_MIN_FLOP or a dim < FFH_BF16_MIN_DIM): the chain is then that mode's per-layer calls; inner activations NONE / RELU (top layer also SIGMOID; forward also GELU); backward: rows of y, dy, dx and
 * w 16-byte aligned and in_dim % 4 == 0 for every layer whose data gradient is produced.  Anything else returns FFH_ERR_UNSUPPORTED
 * with nothing launched and the caller makes the per-layer calls. */
#define FFH_CHAIN_MAX_LAYERS 8
#define FFH_CHAIN_MAX_WIDTH  512
typedef struct ffh_chain_layer {
  const float* w; const float* bias;
  float* y; float* dy; float* dw; float* db;
  int64_t ldy, lddy;
  int ldw, in_dim, out_dim, activation;
} ffh_chain_layer;
int ffh_mlp_chain_fwd(ffh_ctx* ctx, const float* x, int64_t ldx, const ffh_chain_layer* layers, int nlayers, int64_t batch, ffh_stream s);
int ffh_mlp_chain_bwd(ffh_ctx* ctx, const float* x, int64_t ldx, float* dx, int64_t lddx, const ffh_chain_layer* layers, int nlayers,
                      int64_t batch, int flags, ffh_stream s);

/* ------------------------------------------------------------------ */
/* Strict lower triangle of the pairwise-dot matrix (MLPerf-DLRM's interaction keeps the 351 products
 * i > j of the 27 x 27 matrix; the reference has no operator for it -- its dot interaction is a TODO,
 * [ref: examples/cpp/DLRM/dlrm.cc:53-54] -- so parity is against torch: Z[:, li, lj] with tril_indices(n, n, -1)) */
/* ------------------------------------------------------------------ */
/* out[b][p] = in[b][i][j], p = i (i - 1) / 2 + j over i > j in row-major order; in [batch][n][n] contiguous,
 * out [batch][out_ld] with out_ld >= n (n - 1) / 2 (it may be a column slice of a concat buffer); 2 <= n <= 64 */
int ffh_tril_fwd(ffh_ctx* ctx, float* out, int64_t out_ld, const float* in, int64_t batch, int n, ffh_stream s);
/* in_grad[b][i][j] += out_grad[b][p] for i > j; the other entries of in_grad are left as they are */
int ffh_tril_bwd(ffh_ctx* ctx, float* in_grad, const float* out_grad, int64_t grad_ld, int64_t batch, int n, ffh_stream s);

/* The whole pairwise-dot interaction as one launch each way (what the chain cat -> reshape -> transpose -> batch_matmul
 * -> tril -> cat computes, [ref: tests/ops/test_harness.py:125-177] + MLPerf's triangle; one wave per sample, Z Z^T on
 * v_mfma_f32_32x32x2_f32 -- csrc/interaction.hip).  z: [batch][nrows][d] (sample stride ldz >= nrows * d; row 0 is the
 * bottom-MLP output, rows 1.. the embedding outputs), 2 <= nrows <= 32.
 *   out[b][0 .. d)                 = z[b][0][:]
 *   out[b][d + i (i - 1) / 2 + j]  = <z[b][i], z[b][j]>   for i > j          (fp32, k ascending within MFMA order: 1e-5) */
int ffh_dot_interaction_fwd(ffh_ctx* ctx, const float* z, int64_t ldz, float* out, int64_t ldo,
                            int64_t batch, int nrows, int d, ffh_stream s);
/* z_grad[b][i][:] += sum_{j != i} g(i, j) z[b][j][:]  (g(i, j) = out_grad[b][d + pair(max, min)]),  row 0 also
 * += out_grad[b][0 .. d).  FFH_DOT_BWD_OVERWRITE: store instead of add (z has no other consumer). */
#define FFH_DOT_BWD_OVERWRITE 1
int ffh_dot_interaction_bwd(ffh_ctx* ctx, const float* z, int64_t ldz, const float* out_grad, int64_t ldg,
                            float* z_grad, int64_t ldzg, int64_t batch, int nrows, int d, int flags, ffh_stream s);

/* ------------------------------------------------------------------ */
/* Loss, metrics, optimizer                                           */
/* ------------------------------------------------------------------ */
/* mean_squared_error_avg_loss_backward + scale_kernel(0, scale)
 * [ref: src/loss_functions/loss_functions.cu:65-76,160-166; src/runtime/cuda_helper.cu:33-40]
 * logit_grad[i] = (logit[i] - label[i]) * scale, scale = 1/global_batch [ref: loss_functions.cu:202] */
int ffh_mse_bwd(ffh_ctx* ctx, float* logit_grad, const float* logit, const float* label,
                int64_t count, float scale, ffh_stream s);
/* compute_metrics + loss backward of FFModel::backward [ref: src/runtime/model.cc:1443-1452] in one
 * launch: exactly ffh_metrics_update followed by ffh_mse_bwd on the same logits/labels. */
int ffh_mse_bwd_metrics(ffh_ctx* ctx, float* logit_grad, const float* logit, const float* label,
                        ffh_perf_metrics* perf, int64_t num_samples, int num_classes, float scale,
                        int flags, ffh_stream s);
/* update_metrics_label_kernel [ref: src/metrics_functions/metrics_functions.cu:108-173], label
 * (dense) form, restricted to accuracy + MSE/RMSE/MAE.  flags: bit0 accuracy, bit1 mse, bit2 rmse,
 * bit3 mae.  Accumulates into *perf (device memory).  With num_classes == 1 and accuracy on,
 * train_all is counted twice per sample, as the reference does (:119-125). */
int ffh_metrics_update(ffh_ctx* ctx, const float* logits, const float* labels, ffh_perf_metrics* perf,
                       int64_t num_samples, int num_classes, int flags, ffh_stream s);
/* sgd_update [ref: src/runtime/optimizer_kernel.cu:23-41]; v may be NULL when momentum == 0 */
int ffh_sgd_update(ffh_ctx* ctx, float* w, const float* w_grad, float* v, int64_t count,
                   float lr, float weight_decay, float momentum, int nesterov, ffh_stream s);
/* The same update with flags.  FFH_OPT_ZERO_GRAD: w_grad is cleared by the kernel that consumed it, which
 * replaces the next step's Op::zero_grad sweep over it [ref: src/runtime/model.cc:466-490] (the shim keeps
 * track of which gradient buffers are already clean). */
#define FFH_OPT_ZERO_GRAD 1
int ffh_sgd_update_ex(ffh_ctx* ctx, float* w, float* w_grad, float* v, int64_t count,
                      float lr, float weight_decay, float momentum, int nesterov, int flags, ffh_stream s);
/* adam_update [ref: src/runtime/optimizer_kernel.cu:206-226]:
 *   gt = g + weight_decay*w;  m = beta1*m + (1-beta1)*gt;  v = beta2*v + (1-beta2)*gt*gt;
 *   w -= alpha_t * m / (sqrt(v) + epsilon)
 * alpha_t carries the bias correction and is advanced by the caller (AdamOptimizer::next,
 * [ref: src/runtime/optimizer.cc:248-254]).  Canonical rounding (so that the HIP kernel and the oracle agree
 * bit for bit): every a*b+c above is one fused multiply-add, sqrt and the division are correctly rounded. */
int ffh_adam_update(ffh_ctx* ctx, float* w, float* w_grad, float* m, float* v, int64_t count,
                    float alpha_t, float beta1, float beta2, float weight_decay, float epsilon, int flags, ffh_stream s);
/* apply_add_with_scale [ref: src/runtime/cuda_helper.cu:99-108] : dst += src*scale */
int ffh_add_scaled(ffh_ctx* ctx, float* dst, const float* src, int64_t count, float scale, ffh_stream s);
/* dst[i] = src[i] + src[stride + i] + ... + src[(nslices - 1) * stride + i], i < count, added in that order (fp32; dst may be src's slice 0).
 * The local step of the DIRECT all-reduce of the MLP gradients (host: --direct-allreduce): every rank receives one slice of the bucket from
 * every rank over all links at once (all-to-all), sums them in RANK order with this kernel -- every rank computes the same bits -- and the sums
 * are gathered back.  Replaces ncclAllReduce's ring, which one xGMI link bounds [ref: src/runtime/optimizer_kernel.cu:170-171]. */
int ffh_sum_slices_f32(ffh_ctx* ctx, float* dst, const float* src, int nslices, int64_t count, int64_t stride, ffh_stream s);

#ifdef __cplusplus
}
#endif

/* X-macro list of every exported symbol (used by the dlopen loader of the C++
 * FFModel shim and by tests/test_abi_symbols.py). */
#define FFH_API_LIST(X) \
  X(ffh_abi_version) X(ffh_backend_name) X(ffh_ctx_create) X(ffh_ctx_destroy) X(ffh_ctx_default) \
  X(ffh_last_error_string) X(ffh_device_query) X(ffh_ctx_set_workspace) X(ffh_ctx_set_math_mode) X(ffh_ctx_set_deterministic) X(ffh_ctx_set_dw_cu_reserve) X(ffh_ctx_reserve_scratch) X(ffh_ctx_bf16_mirror_set) X(ffh_convert_f32_to_bf16) X(ffh_ctx_bf16x3_mirror_set) X(ffh_convert_f32_to_bf16x3) \
  X(ffh_malloc) X(ffh_free) X(ffh_memcpy_h2d) X(ffh_memcpy_d2h) X(ffh_memcpy_d2d) \
  X(ffh_stream_create) X(ffh_stream_create_with_priority) X(ffh_stream_destroy) X(ffh_stream_sync) X(ffh_device_sync) \
  X(ffh_event_create) X(ffh_event_create_sync) X(ffh_event_destroy) X(ffh_event_record) X(ffh_event_sync) \
  X(ffh_stream_wait_event) X(ffh_event_elapsed_ms) \
  X(ffh_graph_begin_capture) X(ffh_graph_end_capture) X(ffh_graph_launch) X(ffh_graph_destroy) \
  X(ffh_fill_f32) X(ffh_zero) X(ffh_init_uniform) \
  X(ffh_gen_indices) X(ffh_gen_uniform01) X(ffh_gen_bernoulli) \
  X(ffh_embedding_fwd) X(ffh_embedding_fwd_multi) X(ffh_embedding_bwd_dense) \
  X(ffh_embedding_bwd_sgd_fused) X(ffh_embedding_bwd_sgd_fused_multi) X(ffh_embedding_bwd_sort_multi) X(ffh_embedding_bwd_sgd_apply_multi) X(ffh_embedding_bwd_opt_fused_multi) X(ffh_embedding_bwd_opt_apply_multi) \
  X(ffh_embedding_bwd_workspace_bytes) X(ffh_embedding_localize_rows) \
  X(ffh_linear_fwd) X(ffh_linear_fast_in_dim) X(ffh_linear_last_route) X(ffh_embedding_last_route) X(ffh_linear_bwd) X(ffh_linear_bwd_ex) X(ffh_linear_bwd_mse) X(ffh_linear_pair_bwd) X(ffh_linear_pair_fwd) X(ffh_mlp_chain_fwd) X(ffh_mlp_chain_bwd) X(ffh_second_stream_used) X(ffh_event_record_with_next_linear_bwd) X(ffh_linear_bwd_set_dx_scatter) X(ffh_linear_dx_scatter_used) X(ffh_linear_bwd_set_dx_colsum) X(ffh_linear_dx_colsum_used) X(ffh_concat_fwd) X(ffh_concat_bwd) X(ffh_concat_bwd_ex) \
  X(ffh_bmm_fwd) X(ffh_bmm_bwd) X(ffh_transpose_fwd) X(ffh_transpose_bwd) X(ffh_tril_fwd) X(ffh_tril_bwd) X(ffh_dot_interaction_fwd) X(ffh_dot_interaction_bwd) X(ffh_mse_bwd) X(ffh_mse_bwd_metrics) X(ffh_metrics_update) \
  X(ffh_sgd_update) X(ffh_sgd_update_ex) X(ffh_adam_update) X(ffh_add_scaled) X(ffh_sum_slices_f32)

#endif /* FF_HIP_H_ */
