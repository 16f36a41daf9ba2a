// ffh_common.h -- internals shared by the HIP translation units of libffhip.so
// (gfx950 / MI355X only; wave = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS per CU).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ff_hip.h"
#include "../../include/ffh_rng.h"

struct ffh_mirror_region { const char* base; size_t bytes; char* twin; int planes; };   // ffh_ctx_bf16_mirror_set (planes 1: bf16 twin), ffh_ctx_bf16x3_mirror_set (planes 3: I32 image)

struct ffh_ctx {
  int         device;
  void*       ws;        // caller-attached scratch (FFHandler.workSpace analogue)
  size_t      ws_bytes;
  int         num_cus;
  hipEvent_t  ev_fork;   // ffh_linear_bwd_ex: orders the weight-gradient stream behind the caller's stream
  int         second_stream_used;   // ffh_second_stream_used()
  const void* scatter_map;          // ffh_linear_bwd_set_dx_scatter(): ffh_col_dest[scatter_ncols] in device memory, or NULL
  int         scatter_ncols, scatter_used;
  void*       scatter_event;        // hipEvent_t attached to the launch that takes the map (else dropped)
  float*      colsum_dst;           // ffh_linear_bwd_set_dx_colsum(): pending for the next ffh_linear_bwd*, or NULL
  int         colsum_ncols, colsum_used;
  void*       attach_event;         // ffh_event_record_with_next_linear_bwd(): hipEvent_t to signal behind the next backward's last kernel
  int         deterministic;   // ffh_ctx_set_deterministic(): no fp atomics in weight / bias gradients
  int         dw_cu_reserve;   // ffh_ctx_set_dw_cu_reserve(): CUs the persistent weight-gradient GEMMs leave free
  int         math_mode; // ffh_ctx_set_math_mode(): FFH_MATH_DEFAULT | FFH_MATH_TENSOR_OP_BF16
  float*      zeros;     // 256 zero bytes in device memory: source of out-of-range LDS-DMA chunks (linear.hip)
  ffh_mirror_region mirrors[64];   // bf16 twins (tensor-op mode) and three-plane images (split mode) of fp32 buffers
  int         nmirrors;
  // ctx-owned scratch per stream, reserved by ffh_ctx_reserve_scratch(ctx, stream) and released by ffh_stream_destroy / ffh_ctx_destroy;
  // no compute entry point allocates:
  //   sk_slots / sk_cnt    stream-K partial tiles of the persistent fp32 GEMMs (linear_sk.hip): 2 x num_cus slots of 128 x 128 floats + one
  //                        arrival counter per range
  //   skinny_ws / _cnt     one-launch narrow-layer backward (linear_skinny_bwd_kernel): the workgroups' partial dW / db rows + an arrival counter
  //   x3_slots             split mode, weight gradient from images (linear_x3_dma.hip): kX3DwSlots partial tiles of 256 x 256 floats, one per (tile, k-slice)
  struct { void* stream; float* sk_slots; unsigned* sk_cnt; float* skinny_ws; unsigned* skinny_cnt; float* x3_slots; } scratch[FFH_MAX_SCRATCH_STREAMS];
  int         nscratch;
  const void* emb_sorted_ws;     // ffh_embedding_bwd_sort_multi left a sorted list (and cleared fold counters) in THIS workspace ...
  int64_t     emb_sorted_sig[4]; // ... for this (ntables, in_dim, out_dim, batch): ffh_embedding_bwd_sgd_apply_multi consumes it, once
  char        route[256]; // ffh_linear_last_route(): kernel families of the latest ffh_linear_* call
  char        emb_route[64]; // ffh_embedding_last_route(): the form the latest fused table update took
  char        err[512];
};

// ffh_linear_last_route bookkeeping: every ffh_linear_* entry clears the note, every GEMM launch site appends a token
static inline void ffh_route_clear(ffh_ctx* c) { if (c) c->route[0] = 0; }
static inline void ffh_route_add(ffh_ctx* c, const char* token) {
  if (!c) return;
  const size_t n = strlen(c->route);
  snprintf(c->route + n, sizeof c->route - n, "%s%s", n ? ";" : "", token);
}

static inline int ffh_fail(ffh_ctx* c, int code, const char* msg) {
  if (c) snprintf(c->err, sizeof c->err, "%s", msg);
  return code;
}

static inline int ffh_fail_hip(ffh_ctx* c, hipError_t e, const char* what) {
  if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
  return FFH_ERR_HIP;
}

#define FFH_HIP_TRY(ctx, expr)                                  \
  do {                                                          \
    hipError_t e__ = (expr);                                    \
    if (e__ != hipSuccess) return ffh_fail_hip((ctx), e__, #expr); \
  } while (0)

// after a kernel launch: surface launch-configuration errors without synchronising
#define FFH_LAUNCH_CHECK(ctx, name)                             \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return ffh_fail_hip((ctx), e__, name); \
  } while (0)

#define FFH_REQUIRE(ctx, cond, msg)                             \
  do {                                                          \
    if (!(cond)) return ffh_fail((ctx), FFH_ERR_BAD_ARG, msg);  \
  } while (0)

// sizes of the per-stream scratch sets (ffh_ctx_reserve_scratch, runtime.hip)
constexpr int kSkinnyWsBlocks = 512;                       // workgroups the narrow-layer backward's scratch holds
constexpr int kSkinnyWsRow = 4160;                          // floats per partial row: out * in + out for out <= 4, in <= 1024 (4100) and out <= 16, in <= 256 (4112)
constexpr int kSkTileFloats = 128 * 128;                    // one stream-K slot (linear_sk.hip's tile)
constexpr int kX3DwSlots = 512;                             // 134 MB: 3456 x 1024 at 32768 samples is 56 tiles x 9 k-slices
constexpr int kX3TileFloats = 256 * 256;
static inline hipStream_t as_stream(ffh_stream s) { return (hipStream_t)s; }

// A/B switches of the development builds.  The release library reads NO environment variable ("no global state" behind the
// C-ABI): every switch below is its default, a compile-time constant.  tools/build_variant.sh <out.so> -DFFH_LAB builds a lab library
// whose switches come from the environment (tools/ab.sh, tools/gemm_tune.py, tools/gemm_big.py: FFH_TOOLS_LIB=<out.so>).
#ifdef FFH_LAB
#include <stdlib.h>
#define FFH_LAB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define FFH_LAB_I64(name, dflt) (getenv(name) ? atoll(getenv(name)) : (dflt))
#define FFH_LAB_F64(name, dflt) (getenv(name) ? atof(getenv(name)) : (dflt))
#else
#define FFH_LAB_INT(name, dflt) (dflt)
#define FFH_LAB_I64(name, dflt) (dflt)
#define FFH_LAB_F64(name, dflt) (dflt)
#endif

static inline bool ffh_split_mode(const ffh_ctx* c) { return c->math_mode == FFH_MATH_FP32_SPLIT_BF16X3 || c->math_mode == FFH_MATH_FP32_SPLIT_BF16X3_ALL; }

// the bf16 twin of the fp32 element at p when [p, p + span_bytes) lies inside a registered region and the tensor-op mode is on
static inline unsigned short* ffh_mirror_of(const ffh_ctx* c, const void* p, size_t span_bytes) {
  if (!c || !p || c->math_mode != FFH_MATH_TENSOR_OP_BF16) return nullptr;
  const char* q = (const char*)p;
  for (int i = 0; i < c->nmirrors; i++) {
    const ffh_mirror_region& r = c->mirrors[i];
    if (r.planes == 1 && q >= r.base && q + span_bytes <= r.base + r.bytes) return (unsigned short*)(r.twin + (q - r.base) / 2);
  }
  return nullptr;
}

// The three-plane image (ff_hip.h, "I32") of the fp32 element at p: the address of the 192-byte group that holds it, when [p, p + span_bytes)
// lies inside a region registered with ffh_ctx_bf16x3_mirror_set and (any_mode or the split mode is on).  *col0 receives the element's
// position inside its group (0..31); with col0 == nullptr the element must start a group.
constexpr int kMirrorRegions = 64;
static inline char* ffh_planes_of(const ffh_ctx* c, const void* p, size_t span_bytes, int* col0 = nullptr, bool any_mode = false) {
  if (!c || !p || (!any_mode && !ffh_split_mode(c))) return nullptr;
  const char* q = (const char*)p;
  for (int i = 0; i < c->nmirrors; i++) {
    const ffh_mirror_region& r = c->mirrors[i];
    if (r.planes != 3 || q < r.base || q + span_bytes > r.base + r.bytes) continue;
    const size_t e = (size_t)(q - r.base) / 4;
    if (col0) *col0 = (int)(e & 31); else if (e & 31) return nullptr;
    return r.twin + (e >> 5) * 192;
  }
  return nullptr;
}
// byte offset of term 0 of column c (counted from a group's first element) inside a row of the image
__host__ __device__ static inline int64_t ffh_i32_off(int64_t c) { return (c >> 5) * 192 + (c & 31) * 2; }

// grid sizing for memory-bound grid-stride kernels: enough workgroups to fill
// 256 CUs x 8 blocks, capped (cdna guide, Guideline 11)
static inline unsigned ffh_grid(int64_t work_items, int per_block, unsigned cap = 2048) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > (int64_t)cap) g = cap;
  return (unsigned)g;
}

constexpr int kWave = 64;

#ifdef __HIPCC__
// Wave priority of every kernel except the persistent GEMMs (linear_sk.hip): s_setprio 3 at kernel entry.  A persistent workgroup keeps
// one wave per SIMD issuing MFMAs back to back for hundreds of microseconds; the waves of the short kernels that share those CUs (the
// bottom MLP's backward, the table update, the gather) are the younger ones at the issue arbiter and run many times slower than alone.
// Round 4 measured the priority twice: with the bias sums still inside the weight-gradient GEMMs it changed the step by nothing (4096
// samples: the bottom MLP's first dX kernel 210 -> 145 us, step 1.199 -> 1.20-1.22 ms) and stayed off; once those GEMMs ran without the sums
// (ABI 10) it paid a little (32768 samples 7.87 -> 7.83-7.85 ms, 4096: 1.190-1.195 -> 1.176-1.187, MLPerf shape level), and with
// the embedding stream at a higher HIP priority 7.78-7.84 (profiles/r04_ab_schedule.txt).  -DFFH_PRIO=0 builds it out.
#ifndef FFH_PRIO
#define FFH_PRIO 3
#endif
__device__ __forceinline__ void ffh_kernel_prio() {
#if FFH_PRIO > 0
  __builtin_amdgcn_s_setprio(FFH_PRIO);
#endif
}

// 16-byte global access at 4-byte alignment: gfx950 (unaligned-access mode, the amdhsa default) serves a dwordx4 load /
// store at any dword address, so leading dimensions that are not multiples of 4 floats keep the wide accesses
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4u(const float* p) {
  const float4u v = *reinterpret_cast<const float4u*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4u(float* p, const float4 v) {
  float4u u; u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
  *reinterpret_cast<float4u*>(p) = u;
}

// x -> (x1, x2, x3), three bfloat16 terms of an fp32 value (FFH_MATH_FP32_SPLIT_BF16X3, ff_hip.h): x1 = bf16(x) nearest-even, x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2); both differences are exact in fp32.  Four elements at a time, 22 VALU instructions.  The unpack and the subtraction
// are spelled as instructions: left to itself hipcc re-converts the low element (v_cvt_pk_bf16_f32 + shift instead of a shift of the packed
// word) and packs the subtractions into v_pk_add_f32, which costs four times a v_sub_f32 beside an MFMA.  (An infinite x gives x - x1 = NaN.)
typedef __bf16 ffh_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 ffh_pack_bf16x4(const float4 v) {
  const ffh_bf16x2 lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}
__device__ __forceinline__ float ffh_bf16_lo_as_f32(const unsigned p) { unsigned r; asm("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float ffh_bf16_hi_as_f32(const unsigned p) { unsigned r; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float ffh_sub_f32(const float a, const float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float4 ffh_residual_f32x4(const float4 v, const uint2 p) {
  return make_float4(ffh_sub_f32(v.x, ffh_bf16_lo_as_f32(p.x)), ffh_sub_f32(v.y, ffh_bf16_hi_as_f32(p.x)), ffh_sub_f32(v.z, ffh_bf16_lo_as_f32(p.y)), ffh_sub_f32(v.w, ffh_bf16_hi_as_f32(p.y)));
}
__device__ __forceinline__ void ffh_split_bf16x3(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
  p1 = ffh_pack_bf16x4(v);
  const float4 r = ffh_residual_f32x4(v, p1);
  p2 = ffh_pack_bf16x4(r);
  p3 = ffh_pack_bf16x4(ffh_residual_f32x4(r, p2));
}

#endif
