// linear_gemm.h -- what the translation units of the Linear kernels share: the GEMM argument block, the epilogue / guarded-load
// helpers and the entry point of the bf16-pipe GEMMs (linear_bf16.hip), which linear.hip calls for the two bf16 math modes.
#pragma once
#include "ffh_common.h"

#include <hip/hip_runtime.h>

namespace ffh_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_ADD = 1, EPI_ATOMIC = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  float*       C;
  const float* bias;
  int64_t sAm, sAk, sBn, sBk, ldc;
  int64_t bsA, bsB, bsC;     // batch strides (grid.z = batch when splitk == 1)
  int M, N, K;
  int k_per_split;           // multiple of kSplitGran; grid.z = split when splitk > 1
  int splitk;
  int epi;
  int act;
  // dW form only (FUSE_DY): relu'(y) applied to the dy operand as it is loaded (and written back in
  // place by the first column of workgroups), bias gradient = column sums of the same tiles
  const float* act_y;
  int64_t      ld_act_y;
  float*       db;
  int          fuse;         // bit0: relu mask from act_y, bit1: db += column sums
  // FFH_LINEAR_DX_MASK_BY_X: C = mask[m][n] > 0 ? v : 0 in the epilogue
  const float* mask;
  int64_t      ldmask;
  // CMAP kernels (dX of the layer above a Concat, ffh_linear_bwd_set_dx_scatter): column n of C lives at colmap[n].base[m * colmap[n].ld]
  const ffh_col_dest* colmap;
  // persistent dX kernel, plain store epilogue (ffh_linear_bwd_set_dx_colsum): colsum[n] += sum_m C[m][n] as stored, or null
  float* colsum;
  // tensor-op mode with bf16 twins (ffh_ctx_bf16_mirror_set): the operands' twins (same element strides; both or neither) and
  // the twin the epilogue writes beside C (or null)
  const unsigned short* A16;
  const unsigned short* B16;
  unsigned short*       C16;
  int                   a_not_twinned;   // the caller changed A in place without its twin (live activation gradient): do not read A's twin
  int                   db_done;         // out: the bf16-pipe weight-gradient launch also produced db (asked for by db != null, fuse == 0)
  // persistent launch: the (x, y, z) tile space; the launch is then a 1-D grid of fewer workgroups, each walking tiles
  // blockIdx.x, blockIdx.x + gridDim.x, ...  (tnx == 0: one workgroup per tile, tile space = the 3-D grid)
  unsigned tnx, tny, tnz;
};

constexpr int kSplitGran = 32;   // split-K granularity; splits are multiples of 2*kSplitGran = 64 = the largest BK

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == FFH_AC_MODE_RELU) return v > 0.0f ? v : 0.0f;
  if (act == FFH_AC_MODE_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == FFH_AC_MODE_GELU) {     // tanh form, forward only [ref: gelu_forward_kernel, src/runtime/cuda_helper.cu:81-90; src/ops/linear.cu:454-459]
    constexpr float B = 0.7978845608028654f, C = 0.035677408136300125f;
    return v * (0.5f + 0.5f * tanhf(v * (C * v * v + B)));
  }
  return v;
}

// Load 4 consecutive elements along the contiguous dimension (index c0..c0+3 < climit) of
// row `r` (valid if r < rlimit).  p points at element (r, c0).
__device__ __forceinline__ float4 load4_guard(const float* p, bool row_ok, int c0, int climit, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return v;
  if (vec_ok && c0 + 3 < climit) return ld4u(p);
  if (c0 + 0 < climit) v.x = p[0];
  if (c0 + 1 < climit) v.y = p[1];
  if (c0 + 2 < climit) v.z = p[2];
  if (c0 + 3 < climit) v.w = p[3];
  return v;
}

template <typename K>
bool glds_set_lds(K kern, int bytes) {
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

inline bool use_bf16(const ffh_ctx* c, int in, int out, int64_t batch) {     // either bf16-pipe mode: same launch paths
  if (in < FFH_BF16_MIN_DIM || out < FFH_BF16_MIN_DIM) return false;
  if (c->math_mode == FFH_MATH_TENSOR_OP_BF16) return true;
  // the split mode only where it is the faster exact-to-fp32 form -- GEMMs of at least FFH_BF16X3_MIN_FLOP: at 32768 samples 512 -> 256 (8.6 GFLOP)
  // forward takes 88-111 us on the split kernels against 66-73 us on the fp32 ones, 256 -> 128 70 against 22, 1024 -> 512 (34 GFLOP) 153 against 243;
  // at 4096 samples 1024 -> 512 (4.3 GFLOP) 65 against 48, 1024 -> 1024 (8.6) 65 against 78, 3456 -> 1024 (29) 192 against 232
  return c->math_mode == FFH_MATH_FP32_SPLIT_BF16X3_ALL || (c->math_mode == FFH_MATH_FP32_SPLIT_BF16X3 && 2.0 * (double)batch * (double)in * (double)out >= FFH_BF16X3_MIN_FLOP);
}

// The GEMM forms of a Linear layer on the bf16 matrix pipe (tensor-op and fp32-accurate split modes; linear_bf16.hip)
enum { BF16_FORM_FWD = 0,        // A = x (k-contiguous), B = w (k-contiguous)
       BF16_FORM_DW = 1,         // A = dy, B = x, both rows-are-k, split-K with atomics
       BF16_FORM_DX = 2,         // A = dy (k-contiguous), B = w (rows-are-k)
       BF16_FORM_DX_MASK = 3 };  // ... dy read through relu'(act_y)
int launch_gemm_bf16_form(ffh_ctx* c, GemmArgs& g, int form, ffh_stream s, const char* name);
// Tensor-op mode, operands with bf16 twins, outputs of at least one 256 x 256 tile per CU (linear_bf16_dma.hip): LDS-DMA operand
// path, two wave groups alternating on the matrix pipe.  g.A16 / g.B16 / g.C16 as launch_gemm_bf16_form found them.
int launch_gemm_bf16_dma(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name);     // 1 launched, 0 not served, < 0 error
// Split mode, both operands with three-plane images (ffh_ctx_bf16x3_mirror_set) that start a 32-element group, leading dimensions and
// reduction depth multiples of 32 (linear_x3_dma.hip): the LDS-DMA form of the fp32-accurate GEMM; writes the image of C beside C.
int launch_gemm_x3_dma(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name);       // 1 launched, 0 not served, < 0 error
// The k-slices of a weight gradient on 256 x 256 tiles meeting through the stream's reserved slots + an ordered pass instead of float atomics
// (both LDS-DMA kernels; linear_x3_dma.hip): slice ks of tile t stores its tile at slots + (t * splitk + ks) * 65536, then C[tile] += sum over ks in order
float* dw_tile_slots(const ffh_ctx* c, ffh_stream s, int64_t tiles, int splitk, int64_t ldc);          // null: use atomics
void launch_dw_tile_reduce(const float* slots, float* C, int64_t ldc, int M, int N, int64_t tiles, int splitk, ffh_stream s);

// The persistent one-workgroup-per-CU fp32 kernels for the big aligned layers (linear_sk.hip): forward (bias + activation),
// data gradient (store / add, optional relu'-of-the-layer-below mask), weight gradient (stream-K, atomics).
enum { SK_FORM_FWD = 0, SK_FORM_DX = 1, SK_FORM_DW = 2 };
bool gemm_sk_serves(const ffh_ctx* c, const GemmArgs& g, int form);                                   // shape / alignment / mode check only
int launch_gemm_sk(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name);           // 1 launched, 0 not served, < 0 error

}  // namespace ffh_gemm
