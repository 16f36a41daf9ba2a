// linear_bf16.hip -- the Linear GEMMs on the bf16 matrix pipe: tensor-op math mode (bf16 operands, fp32 accumulate) and the
// fp32-accurate three-way split.  Its own translation unit: the tile-shape instantiations compile beside linear.hip.
#include "linear_gemm.h"

#include <stdlib.h>
#include <type_traits>

using namespace ffh_gemm;

namespace {

// =============================================================================================
// Tensor-op math mode (ffh_ctx_set_math_mode(FFH_MATH_TENSOR_OP_BF16); the reference's
// --allow-tensor-op-math-conversion -> cublasSetMathMode(CUBLAS_TENSOR_OP_MATH) [ref: src/runtime/model.cu:81-83]):
// the same three GEMM forms with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 MFMA rate).
// Activations, weights and gradients stay fp32 in HBM; a tile is rounded to bf16 (v_cvt_pk_bf16_f32, nearest even)
// between its global load and its LDS image, so nothing outside this kernel knows about the mode.
//   128 x 128 x 64 block tile, 4 waves (2 x 2), each 64 x 64 = 2 x 2 MFMA tiles; double-buffered LDS (64 KB: two
//   workgroups per CU), one barrier per k-tile, global loads of tile t+1 in flight under the MFMAs of tile t.
//   k-contiguous operand (x, w forward; dy in dX): image [row][64 bf16] = 128-byte rows, 16-byte chunks XOR-swizzled by
//     (row >> 1) & 7 -> the fragment of lane (row r, half h) at k-step s is ONE conflict-free ds_read_b128 (chunk 2s + h);
//   rows-are-k operand (w in dX; dy, x in dW): image [k][128 bf16] = 256-byte rows, stored as it arrives (one ds_write_b64
//     per float4, coalesced), read TRANSPOSED by ds_read_b64_tr_b16: a 16-lane group fetches a 4 (k) x 16 (column) block and
//     every lane receives the 4 k-values of its own column; two such reads make the 8-element MFMA fragment.  Chunk swizzle
//     ch ^ (((k & 3) << 2) | ((k >> 2) & 3)) keeps both the stores and the transposed reads conflict-free.
// With fp32 operands in memory the kernel is bound by the global -> LDS path (32 flop per staged byte), not by the matrix
// pipe: ~0.8-1 PFLOP/s, i.e. 6-8x the fp32 mode.
// =============================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 pack_bf16x4(const float4 v) {
  const bf16x2_t lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}

constexpr int kBfBM = 128, kBfBN = 128, kBfBK = 64;
constexpr int kBfImage = 128 * 64 * 2;                    // bytes of one 128-row operand image (either layout)
constexpr int kBfLds = 2 * 2 * kBfImage;                  // two operands, two buffers
constexpr int bf_lds_bytes(int bm, int bn) { return 2 * (bm + bn) * kBfBK * 2; }

__device__ __forceinline__ unsigned bf_off_kc(int row, int chunk) { return (unsigned)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); }
// rows-are-k image of an operand with ROWLEN (128 or 256) columns: 2 * ROWLEN bytes per k-row, the swizzle acts on the low four chunk bits
template <int ROWLEN = 128>
__device__ __forceinline__ unsigned bf_off_kr(int krow, int chunk) { return (unsigned)(krow * (2 * ROWLEN) + ((chunk ^ (((krow & 3) << 2) | ((krow >> 2) & 3))) << 4)); }

// MASK_A: the A operand (k-contiguous dy of the dX form) is read through relu'(act_y) (FFH_LINEAR_ONLY_DX / forked dW)
// BM x BN block tile, one wave per WM x 64 of it: 128 x 128 (4 waves of 64 x 64, two workgroups per CU) or, where the output
// has enough tiles to fill the chip with them, 256 x 256 (8 waves of 128 x 64, 128 KB of LDS, one workgroup per CU): the fp32
// operands cross the L2 -> CU path half as often, and that path is what bounds this kernel
// SRC16: both operands come from their bf16 twins (ffh_ctx_bf16_mirror_set; K a multiple of 64, M, N and the strides multiples
// of 8): a thread moves 16-byte chunks of 8 elements global -> LDS with no conversion and half the bytes
template <bool AKC, bool BKC, bool MASK_A = false, int BM = kBfBM, int BN = kBfBN, int WM = 64, bool SRC16 = false>
__global__ __launch_bounds__(BM / WM * BN) void gemm_bf16_kernel(const GemmArgs g) {
  ffh_kernel_prio();
  static_assert(!(SRC16 && MASK_A), "the masking loader reads fp32 operands");
  constexpr int BK = kBfBK, NT = BM / WM * BN;                 // one wave per WM x 64 of the tile
  constexpr int TM = WM / 32;
  constexpr int NA = SRC16 ? BM * 8 / NT : BM * 16 / NT, NB = SRC16 ? BN * 8 / NT : BN * 16 / NT;   // 16-byte pieces per thread per k-tile
  constexpr int IMG_A = BM * BK * 2, IMG_B = BN * BK * 2;     // bytes of the operand images
  static_assert(NT <= 1024 && NA >= 1 && NB >= 1, "tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char bf_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx, by, bz;
  {
    const unsigned nbx = gridDim.x, nby = gridDim.y, nbz = gridDim.z;
    const unsigned total = nbx * nby * nbz;
    const unsigned lin = (blockIdx.z * nby + blockIdx.y) * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx);
    by = (int)((nlin / nbx) % nby);
    bz = (int)(nlin / (nbx * nby));
  }
  const int m0 = by * BM, n0 = bx * BN;
  int kb = 0, ke = g.K;
  if (g.splitk > 1) {
    kb = bz * g.k_per_split;
    ke = kb + g.k_per_split < g.K ? kb + g.k_per_split : g.K;
  }
  if (kb >= ke) return;
  const int nk = (ke - kb + BK - 1) / BK;
  const float* A = g.A;
  const float* B = g.B;

  float4 ra[NA], rb[NB];
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t pa[NA], pb[NB];          // SRC16: 16-byte pieces of the twins, moved as they are
  const bool a_in = m0 + BM <= g.M, b_in = n0 + BN <= g.N;
  auto load_tile = [&](int kt, auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int k0 = kb + kt * BK;
    if constexpr (SRC16) {      // 8 bf16 per piece; a k-contiguous row is 8 pieces, a k-row of the other form BM / 8 (BN / 8)
#pragma unroll
      // rows / columns of an edge tile beyond the matrix are clamped to its last row / last 8 columns (M, N multiples of 8): valid
      // addresses, values the epilogue never stores
      for (int i = 0; i < NA; i++) {
        if (AKC) { const int ch = tid & 7, row = (tid >> 3) + (NT / 8) * i; const int m = m0 + row < g.M ? m0 + row : g.M - 1; pa[i] = *reinterpret_cast<const u32x4_t*>(g.A16 + (int64_t)m * g.sAm + k0 + 8 * ch); }
        else { const int ch = tid % (BM / 8), kr = tid / (BM / 8) + (8 * NT / BM) * i; const int m = m0 + 8 * ch < g.M ? m0 + 8 * ch : g.M - 8; pa[i] = *reinterpret_cast<const u32x4_t*>(g.A16 + (int64_t)(k0 + kr) * g.sAk + m); }
      }
#pragma unroll
      for (int i = 0; i < NB; i++) {
        if (BKC) { const int ch = tid & 7, row = (tid >> 3) + (NT / 8) * i; const int n = n0 + row < g.N ? n0 + row : g.N - 1; pb[i] = *reinterpret_cast<const u32x4_t*>(g.B16 + (int64_t)n * g.sBn + k0 + 8 * ch); }
        else { const int ch = tid % (BN / 8), kr = tid / (BN / 8) + (8 * NT / BN) * i; const int n = n0 + 8 * ch < g.N ? n0 + 8 * ch : g.N - 8; pb[i] = *reinterpret_cast<const u32x4_t*>(g.B16 + (int64_t)(k0 + kr) * g.sBk + n); }
      }
    } else {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {
        const int k4 = tid & 15, row = (tid >> 4) + (NT / 16) * i;
        const int m = m0 + row, k = k0 + 4 * k4;
        if (FAST) ra[i] = ld4u(A + (int64_t)m * g.sAm + k);
        else ra[i] = load4_guard(A + (int64_t)m * g.sAm + k, m < g.M, k, ke, true);
        if (MASK_A) {
          float4 yv;
          if (FAST) yv = ld4u(g.act_y + (int64_t)m * g.ld_act_y + k);
          else yv = load4_guard(g.act_y + (int64_t)m * g.ld_act_y + k, m < g.M, k, ke, true);
          ra[i].x = yv.x > 0.0f ? ra[i].x : 0.0f; ra[i].y = yv.y > 0.0f ? ra[i].y : 0.0f;
          ra[i].z = yv.z > 0.0f ? ra[i].z : 0.0f; ra[i].w = yv.w > 0.0f ? ra[i].w : 0.0f;
        }
      } else {
        const int m4 = tid % (BM / 4), kr = tid / (BM / 4) + (4 * NT / BM) * i;
        const int m = m0 + 4 * m4, k = k0 + kr;
        if (FAST) ra[i] = ld4u(A + (int64_t)k * g.sAk + m);
        else ra[i] = load4_guard(A + (int64_t)k * g.sAk + m, k < ke, m, g.M, true);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid & 15, row = (tid >> 4) + (NT / 16) * i;
        const int n = n0 + row, k = k0 + 4 * k4;
        if (FAST) rb[i] = ld4u(B + (int64_t)n * g.sBn + k);
        else rb[i] = load4_guard(B + (int64_t)n * g.sBn + k, n < g.N, k, ke, true);
      } else {
        const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;
        const int n = n0 + 4 * n4, k = k0 + kr;
        if (FAST) rb[i] = ld4u(B + (int64_t)k * g.sBk + n);
        else rb[i] = load4_guard(B + (int64_t)k * g.sBk + n, k < ke, n, g.N, true);
      }
    }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    unsigned char* as = bf_smem + buf * (IMG_A + IMG_B);
    unsigned char* bs = as + IMG_A;
    if constexpr (SRC16) {      // the same images, one whole 16-byte chunk per store
#pragma unroll
      for (int i = 0; i < NA; i++) {
        if (AKC) { const int ch = tid & 7, row = (tid >> 3) + (NT / 8) * i; *reinterpret_cast<u32x4_t*>(as + bf_off_kc(row, ch)) = pa[i]; }
        else { const int ch = tid % (BM / 8), kr = tid / (BM / 8) + (8 * NT / BM) * i; *reinterpret_cast<u32x4_t*>(as + bf_off_kr<BM>(kr, ch)) = pa[i]; }
      }
#pragma unroll
      for (int i = 0; i < NB; i++) {
        if (BKC) { const int ch = tid & 7, row = (tid >> 3) + (NT / 8) * i; *reinterpret_cast<u32x4_t*>(bs + bf_off_kc(row, ch)) = pb[i]; }
        else { const int ch = tid % (BN / 8), kr = tid / (BN / 8) + (8 * NT / BN) * i; *reinterpret_cast<u32x4_t*>(bs + bf_off_kr<BN>(kr, ch)) = pb[i]; }
      }
    } else {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {
        const int k4 = tid & 15, row = (tid >> 4) + (NT / 16) * i;
        *reinterpret_cast<uint2*>(as + bf_off_kc(row, k4 >> 1) + 8 * (k4 & 1)) = pack_bf16x4(ra[i]);
      } else {
        const int m4 = tid % (BM / 4), kr = tid / (BM / 4) + (4 * NT / BM) * i;
        *reinterpret_cast<uint2*>(as + bf_off_kr<BM>(kr, m4 >> 1) + 8 * (m4 & 1)) = pack_bf16x4(ra[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid & 15, row = (tid >> 4) + (NT / 16) * i;
        *reinterpret_cast<uint2*>(bs + bf_off_kc(row, k4 >> 1) + 8 * (k4 & 1)) = pack_bf16x4(rb[i]);
      } else {
        const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;
        *reinterpret_cast<uint2*>(bs + bf_off_kr<BN>(kr, n4 >> 1) + 8 * (n4 & 1)) = pack_bf16x4(rb[i]);
      }
    }
    }
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  const int wm0 = (wave / (BN / 64)) * WM, wn0 = (wave % (BN / 64)) * 64;
  const int lr = lane & 31, lh = lane >> 5;
  // transposed-read lane roles: 16-lane group tg, row tq and column quad tp of the 4 x 16 block
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // fragment of the 32-row (or 32-column) tile starting at `o` of an image, k-step s (16 k)
  auto frag = [&](const unsigned char* img, bool kc, auto rowlen_tag, int o, int s) -> bf16x8_t {
    constexpr int RL = decltype(rowlen_tag)::value;
    if (kc) return *reinterpret_cast<const bf16x8_t*>(img + bf_off_kc(o + lr, 2 * s + lh));
    typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
    const int ch = (o >> 3) + 2 * (tg & 1) + (tp >> 1);
    const int k0r = 16 * s + 8 * (tg >> 1) + tq;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<RL>(k0r, ch) + 8 * (tp & 1)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<RL>(k0r + 4, ch) + 8 * (tp & 1)));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto compute_tile = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* as = bf_smem + buf * (IMG_A + IMG_B);
    const unsigned char* bs = as + IMG_A;
#pragma unroll
    for (int s = 0; s < BK / 16; s++) {
      bf16x8_t a[TM], b[2];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = frag(as, AKC, std::integral_constant<int, BM>{}, wm0 + 32 * i, s);
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = frag(bs, BKC, std::integral_constant<int, BN>{}, wn0 + 32 * j, s);
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  // k-tiles fully inside the matrices take the unguarded loader (workgroup-uniform choice)
  const int nfull = (a_in && b_in) ? (ke - kb) / BK : 0;
  if (nfull > 0) load_tile(0, std::true_type{}); else load_tile(0, std::false_type{});
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nk; t++) {
    const int buf = t & 1;
    if (t + 1 < nk) { if (t + 1 < nfull) load_tile(t + 1, std::true_type{}); else load_tile(t + 1, std::false_type{}); }
    compute_tile(buf);
    if (t + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  float* C = g.C;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      if (n >= g.N) continue;
      const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* cp = C + (int64_t)m * g.ldc + n;
        float v = acc[i][j][r];
        if (g.mask && !(g.mask[(int64_t)m * g.ldmask + n] > 0.0f)) v = 0.0f;
        if (g.epi == EPI_STORE) v = act_apply(v + bv, g.act);
        else if (g.epi == EPI_ADD) v = *cp + v;
        if (g.epi == EPI_ATOMIC) { atomicAdd(cp, v); continue; }
        *cp = v;
        if (g.C16) { const __bf16 t = (__bf16)v; g.C16[(int64_t)m * g.ldc + n] = __builtin_bit_cast(unsigned short, t); }   // the twin of C
      }
    }
}

// =============================================================================================
// fp32 GEMM on the bf16 matrix pipe: FFH_MATH_FP32_SPLIT_BF16X3.  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate, so an
// fp32-ACCURATE product can be had faster from bf16 pieces: every operand element is split into three bfloat16 terms
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (both residuals are exact in fp32;
//                                                                                       |x - x1 - x2 - x3| <= 2^-27 |x|)
// and a*b is accumulated from the six products a_i b_j with i + j <= 4 -- each exact in fp32 (8 x 8 significant bits), summed
// in fp32 by v_mfma_f32_32x32x16_bf16, small terms first; the three dropped products are <= 2^-26 |a b|, below fp32's own
// rounding.  Six bf16 MFMAs of 32 cycles per 16 k against eight fp32 MFMAs of 64 cycles: 2.67x the fp32 pipe rate at the
// same result to within the fp32 summation-order bound (the parity tests hold this mode to the SAME 1e-5-of-term-mass bound
// as the exact-fp32 kernels, and compare both against float64).  Same tile / layouts as gemm_bf16_kernel with BK = 32 and
// three planes per operand in a single 48 KB LDS buffer (two workgroups per CU); the split happens between the global load
// and the LDS store.  Not the default: an opt-in math mode.
// What bounds it (measured by ablation, 3456 -> 1024 at batch 32768, 1,212 us): not the 48 MFMAs per k-tile (59 % of the
// SIMD cycles at the 1.9 GHz the chip holds under this load) but the staging path -- fp32 operands from L2 (7.2 GB per
// launch: ~410 us by itself) and 48 KB of ds_write_b64 per k-tile (LDS stores run at ~85 B/clk/CU: ~260 us by itself); the
// split's VALU work costs ~20 us once it is spelled as below.  A wave-specialised variant (four waves multiply, four load /
// split / store into a second or third LDS buffer, three k-tiles of loads in flight) measured 186 / 165 / 174 TFLOP/s
// (fwd / dX / dW) against 191 / 172 / 138 here, and the same 6.52 ms for the whole step: not kept.
// =============================================================================================
constexpr int kX3BK = 32;
constexpr int kX3Plane = 128 * kX3BK * 2;               // bytes of one bf16 plane of one operand image
constexpr int kX3Lds = 2 * 3 * kX3Plane;                // two operands x three planes, single buffer

__device__ __forceinline__ unsigned x3_off_kc(int row, int chunk) { return (unsigned)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)); }

// The unpack and the subtraction are spelled as instructions: left to itself hipcc re-converts the low element
// (v_cvt_pk_bf16_f32 + shift instead of a shift of the packed word: 80 conversions per tile where 48 are needed) and packs the
// subtractions into v_pk_add_f32, which costs four times a v_sub_f32 in the shadow of an MFMA.
__device__ __forceinline__ float bf16_lo_as_f32(const unsigned p) { unsigned r; asm("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float bf16_hi_as_f32(const unsigned p) { unsigned r; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float sub_f32(const float a, const float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float4 residual_f32x4(const float4 v, const uint2 p) {
  return make_float4(sub_f32(v.x, bf16_lo_as_f32(p.x)), sub_f32(v.y, bf16_hi_as_f32(p.x)), sub_f32(v.z, bf16_lo_as_f32(p.y)), sub_f32(v.w, bf16_hi_as_f32(p.y)));
}
// x -> (x1, x2, x3): 22 VALU instructions per four elements.  (An infinite x gives x - x1 = NaN: such an operand turns its
// outputs into NaN where fp32 arithmetic gives an infinity -- stated in ff_hip.h.)
__device__ __forceinline__ void split_bf16x3(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
  p1 = pack_bf16x4(v);
  const float4 r = residual_f32x4(v, p1);
  p2 = pack_bf16x4(r);
  p3 = pack_bf16x4(residual_f32x4(r, p2));
}

template <bool AKC, bool BKC, bool MASK_A = false, int BM = 128, int BN = 128, int WM = 64>
__global__ __launch_bounds__(BM / WM * BN) void gemm_bf16x3_kernel(const GemmArgs g) {
  ffh_kernel_prio();
  constexpr int BK = kX3BK, NT = BM / WM * BN;                  // one wave per WM x 64 of the tile
  constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;             // float4 per thread per k-tile
  constexpr int TM = WM / 32;
  constexpr int PLANE_A = BM * BK * 2, PLANE_B = BN * BK * 2;   // bytes of one bf16 plane of an operand image
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx, by, bz;
  {
    const unsigned nbx = gridDim.x, nby = gridDim.y, nbz = gridDim.z;
    const unsigned total = nbx * nby * nbz;
    const unsigned lin = (blockIdx.z * nby + blockIdx.y) * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx);
    by = (int)((nlin / nbx) % nby);
    bz = (int)(nlin / (nbx * nby));
  }
  const int m0 = by * BM, n0 = bx * BN;
  int kb = 0, ke = g.K;
  if (g.splitk > 1) {
    kb = bz * g.k_per_split;
    ke = kb + g.k_per_split < g.K ? kb + g.k_per_split : g.K;
  }
  if (kb >= ke) return;
  const int nk = (ke - kb + BK - 1) / BK;
  const float* A = g.A;
  const float* B = g.B;

  float4 ra0[NA], rb0[NB];
  const bool a_in = m0 + BM <= g.M, b_in = n0 + BN <= g.N;
  auto load_tile = [&](int kt, auto fast_tag, float4 (&ra)[NA], float4 (&rb)[NB]) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int k0 = kb + kt * BK;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {
        const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i;
        const int m = m0 + row, k = k0 + 4 * k4;
        if (FAST) ra[i] = ld4u(A + (int64_t)m * g.sAm + k);
        else ra[i] = load4_guard(A + (int64_t)m * g.sAm + k, m < g.M, k, ke, true);
        if (MASK_A) {
          float4 yv;
          if (FAST) yv = ld4u(g.act_y + (int64_t)m * g.ld_act_y + k);
          else yv = load4_guard(g.act_y + (int64_t)m * g.ld_act_y + k, m < g.M, k, ke, true);
          ra[i].x = yv.x > 0.0f ? ra[i].x : 0.0f; ra[i].y = yv.y > 0.0f ? ra[i].y : 0.0f;
          ra[i].z = yv.z > 0.0f ? ra[i].z : 0.0f; ra[i].w = yv.w > 0.0f ? ra[i].w : 0.0f;
        }
      } else {
        const int m4 = tid % (BM / 4), kr = tid / (BM / 4) + (4 * NT / BM) * i;
        const int m = m0 + 4 * m4, k = k0 + kr;
        if (FAST) ra[i] = ld4u(A + (int64_t)k * g.sAk + m);
        else ra[i] = load4_guard(A + (int64_t)k * g.sAk + m, k < ke, m, g.M, true);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i;
        const int n = n0 + row, k = k0 + 4 * k4;
        if (FAST) rb[i] = ld4u(B + (int64_t)n * g.sBn + k);
        else rb[i] = load4_guard(B + (int64_t)n * g.sBn + k, n < g.N, k, ke, true);
      } else {
        const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;
        const int n = n0 + 4 * n4, k = k0 + kr;
        if (FAST) rb[i] = ld4u(B + (int64_t)k * g.sBk + n);
        else rb[i] = load4_guard(B + (int64_t)k * g.sBk + n, k < ke, n, g.N, true);
      }
    }
  };
  auto split_store = [&](const float4 (&ra)[NA], const float4 (&rb)[NB]) {
    unsigned char* as = x3_smem;
    unsigned char* bs = x3_smem + 3 * PLANE_A;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      uint2 p1, p2, p3;
      split_bf16x3(ra[i], p1, p2, p3);
      unsigned o;
      if (AKC) { const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i; o = x3_off_kc(row, k4 >> 1) + 8 * (k4 & 1); }
      else     { const int m4 = tid % (BM / 4), kr = tid / (BM / 4) + (4 * NT / BM) * i;  o = bf_off_kr<BM>(kr, m4 >> 1) + 8 * (m4 & 1); }
      *reinterpret_cast<uint2*>(as + o) = p1;
      *reinterpret_cast<uint2*>(as + PLANE_A + o) = p2;
      *reinterpret_cast<uint2*>(as + 2 * PLANE_A + o) = p3;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      uint2 p1, p2, p3;
      split_bf16x3(rb[i], p1, p2, p3);
      unsigned o;
      if (BKC) { const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i; o = x3_off_kc(row, k4 >> 1) + 8 * (k4 & 1); }
      else     { const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;  o = bf_off_kr<BN>(kr, n4 >> 1) + 8 * (n4 & 1); }
      *reinterpret_cast<uint2*>(bs + o) = p1;
      *reinterpret_cast<uint2*>(bs + PLANE_B + o) = p2;
      *reinterpret_cast<uint2*>(bs + 2 * PLANE_B + o) = p3;
    }
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  const int wm0 = (wave / (BN / 64)) * WM, wn0 = (wave % (BN / 64)) * 64;
  const int lr = lane & 31, lh = lane >> 5;
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  auto frag = [&](const unsigned char* img, bool kc, auto rowlen_tag, int o, int s) -> bf16x8_t {
    constexpr int RL = decltype(rowlen_tag)::value;
    if (kc) return *reinterpret_cast<const bf16x8_t*>(img + x3_off_kc(o + lr, 2 * s + lh));
    typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
    const int ch = (o >> 3) + 2 * (tg & 1) + (tp >> 1);
    const int k0r = 16 * s + 8 * (tg >> 1) + tq;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<RL>(k0r, ch) + 8 * (tp & 1)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<RL>(k0r + 4, ch) + 8 * (tp & 1)));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto compute_tile = [&]() {
    const unsigned char* as = x3_smem;
    const unsigned char* bs = x3_smem + 3 * PLANE_A;
#pragma unroll
    for (int s = 0; s < BK / 16; s++) {
      bf16x8_t b[2][3];
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int j = 0; j < 2; j++) b[j][p] = frag(bs + p * PLANE_B, BKC, std::integral_constant<int, BN>{}, wn0 + 32 * j, s);
#pragma unroll
      for (int ih = 0; ih < TM; ih += 2) {                  // two row tiles at a time: 6 + 6 fragments live
        bf16x8_t a[2][3];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
          for (int i = 0; i < 2; i++) a[i][p] = frag(as + p * PLANE_A, AKC, std::integral_constant<int, BM>{}, wm0 + 32 * (ih + i), s);
        // the six products with i + j <= 4, small terms first
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[ih + i][j], 0, 0, 0);
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[ih + i][j], 0, 0, 0);
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[ih + i][j], 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[ih + i][j], 0, 0, 0);
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[ih + i][j], 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[ih + i][j], 0, 0, 0);
      }
    }
  };

  const int nfull = (a_in && b_in) ? (ke - kb) / BK : 0;
  auto load_any = [&](int kt, float4 (&ra)[NA], float4 (&rb)[NB]) {
    if (kt >= nk) return;
    if (kt < nfull) load_tile(kt, std::true_type{}, ra, rb); else load_tile(kt, std::false_type{}, ra, rb);
  };
  load_any(0, ra0, rb0);
  split_store(ra0, rb0);
  __syncthreads();
  for (int t = 0; t < nk; t++) {
    load_any(t + 1, ra0, rb0);            // in flight under this tile's MFMAs (a second register set, two tiles ahead, was
    compute_tile();                       //   measured: +2 % forward, -12 % dW -- its 194 VGPRs leave one workgroup per CU)
    __syncthreads();                      // every wave has read this tile's fragments
    if (t + 1 < nk) { split_store(ra0, rb0); __syncthreads(); }
  }

  float* C = g.C;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      if (n >= g.N) continue;
      const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* cp = C + (int64_t)m * g.ldc + n;
        float v = acc[i][j][r];
        if (g.mask && !(g.mask[(int64_t)m * g.ldmask + n] > 0.0f)) v = 0.0f;
        if (g.epi == EPI_STORE) *cp = act_apply(v + bv, g.act);
        else if (g.epi == EPI_ADD) *cp = *cp + v;
        else atomicAdd(cp, v);
      }
    }
}

// =============================================================================================
// The split-in-kernel form of the forward and data-gradient GEMMs with the ARITHMETIC of the LDS-DMA kernel (linear_x3_dma.hip): the same
// instruction (v_mfma_f32_16x16x32_bf16: one instruction per 32-deep k-step and product), the same k-slot assignment (lane group q of a
// fragment holds k = 8q .. 8q + 7 of the step), the same six products in the same order into the same accumulator, k-steps in ascending
// order from a zero accumulator, the same epilogue operations -- so a layer gives the SAME BITS whether its operands come from producer-kept
// images or are split here (tests/test_gpu_round6.py).  Same tiles, staging and LDS images as gemm_bf16x3_kernel above (A is k-contiguous
// in both forms; B k-contiguous forward, rows-are-k in dX); the k-contiguous image's swizzle is the one that makes the 16-row fragments'
// ds_read_b128 conflict-free.  The weight gradient keeps the kernel above: its k-slices meet by atomics, whose order is free anyway.
// =============================================================================================
__device__ __forceinline__ unsigned x3v_off_kc(int row, int chunk) { return (unsigned)(row * 64 + ((chunk ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3)) << 4)); }

template <bool BKC, bool MASK_A = false, int BM = 128, int BN = 128, int WM = 64>
__global__ __launch_bounds__(BM / WM * BN) void gemm_bf16x3_v2_kernel(const GemmArgs g) {
  ffh_kernel_prio();
  constexpr int BK = kX3BK, NT = BM / WM * BN;                  // one wave per WM x 64 of the tile
  constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;             // float4 per thread per k-tile
  constexpr int TM = WM / 16;
  constexpr int PLANE_A = BM * BK * 2, PLANE_B = BN * BK * 2;   // bytes of one bf16 plane of an operand image
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx, by;
  {
    const unsigned nbx = gridDim.x, nby = gridDim.y;
    const unsigned total = nbx * nby;
    const unsigned lin = blockIdx.y * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx);
    by = (int)(nlin / nbx);
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int ke = g.K;
  const int nk = (ke + BK - 1) / BK;
  const float* A = g.A;
  const float* B = g.B;

  float4 ra0[NA], rb0[NB];
  const bool a_in = m0 + BM <= g.M, b_in = n0 + BN <= g.N;
  auto load_tile = [&](int kt, auto fast_tag, float4 (&ra)[NA], float4 (&rb)[NB]) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i;
      const int m = m0 + row, k = k0 + 4 * k4;
      if (FAST) ra[i] = ld4u(A + (int64_t)m * g.sAm + k);
      else ra[i] = load4_guard(A + (int64_t)m * g.sAm + k, m < g.M, k, ke, true);
      if (MASK_A) {
        float4 yv;
        if (FAST) yv = ld4u(g.act_y + (int64_t)m * g.ld_act_y + k);
        else yv = load4_guard(g.act_y + (int64_t)m * g.ld_act_y + k, m < g.M, k, ke, true);
        ra[i].x = yv.x > 0.0f ? ra[i].x : 0.0f; ra[i].y = yv.y > 0.0f ? ra[i].y : 0.0f;
        ra[i].z = yv.z > 0.0f ? ra[i].z : 0.0f; ra[i].w = yv.w > 0.0f ? ra[i].w : 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i;
        const int n = n0 + row, k = k0 + 4 * k4;
        if (FAST) rb[i] = ld4u(B + (int64_t)n * g.sBn + k);
        else rb[i] = load4_guard(B + (int64_t)n * g.sBn + k, n < g.N, k, ke, true);
      } else {
        const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;
        const int n = n0 + 4 * n4, k = k0 + kr;
        if (FAST) rb[i] = ld4u(B + (int64_t)k * g.sBk + n);
        else rb[i] = load4_guard(B + (int64_t)k * g.sBk + n, k < ke, n, g.N, true);
      }
    }
  };
  auto split_store = [&](const float4 (&ra)[NA], const float4 (&rb)[NB]) {
    unsigned char* as = x3_smem;
    unsigned char* bs = x3_smem + 3 * PLANE_A;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      uint2 p1, p2, p3;
      split_bf16x3(ra[i], p1, p2, p3);
      const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i;
      const unsigned o = x3v_off_kc(row, k4 >> 1) + 8 * (k4 & 1);
      *reinterpret_cast<uint2*>(as + o) = p1;
      *reinterpret_cast<uint2*>(as + PLANE_A + o) = p2;
      *reinterpret_cast<uint2*>(as + 2 * PLANE_A + o) = p3;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      uint2 p1, p2, p3;
      split_bf16x3(rb[i], p1, p2, p3);
      unsigned o;
      if (BKC) { const int k4 = tid & 7, row = (tid >> 3) + (NT / 8) * i; o = x3v_off_kc(row, k4 >> 1) + 8 * (k4 & 1); }
      else     { const int n4 = tid % (BN / 4), kr = tid / (BN / 4) + (4 * NT / BN) * i;  o = bf_off_kr<BN>(kr, n4 >> 1) + 8 * (n4 & 1); }
      *reinterpret_cast<uint2*>(bs + o) = p1;
      *reinterpret_cast<uint2*>(bs + PLANE_B + o) = p2;
      *reinterpret_cast<uint2*>(bs + 2 * PLANE_B + o) = p3;
    }
  };

  f32x4_t acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int wm0 = (wave / (BN / 64)) * WM, wn0 = (wave % (BN / 64)) * 64;
  const int c = lane & 15, q = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  // fragment of the 16 rows (columns) starting at `o`: lane (c, q) gets k = 8q .. 8q + 7 of the 32-deep step for row (column) o + c
  auto frag_kc = [&](const unsigned char* img, int o) -> bf16x8_t { return *reinterpret_cast<const bf16x8_t*>(img + x3v_off_kc(o + c, q)); };
  auto frag_kr = [&](const unsigned char* img, int o) -> bf16x8_t {
    const int ch = (o >> 3) + (tp >> 1), r1 = 8 * q + tq;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<BN>(r1, ch) + 8 * (tp & 1)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(img + bf_off_kr<BN>(r1 + 4, ch) + 8 * (tp & 1)));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto compute_tile = [&]() {
    const unsigned char* as = x3_smem;
    const unsigned char* bs = x3_smem + 3 * PLANE_A;
    bf16x8_t b[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int p = 0; p < 3; p++) b[j][p] = BKC ? frag_kc(bs + p * PLANE_B, wn0 + 16 * j) : frag_kr(bs + p * PLANE_B, wn0 + 16 * j);
    constexpr int IB = TM < 4 ? TM : 4;
#pragma unroll
    for (int ih = 0; ih < TM; ih += IB) {                 // (up to) four row blocks at a time: 12 + 12 fragments live
      bf16x8_t a[IB][3];
#pragma unroll
      for (int i = 0; i < IB; i++)
#pragma unroll
        for (int p = 0; p < 3; p++) a[i][p] = frag_kc(as + p * PLANE_A, wm0 + 16 * (ih + i));
      // the six products with i + j <= 4, small terms first -- per accumulator the order of linear_x3_dma.hip
#pragma unroll
      for (int pr = 0; pr < 6; pr++) {
        const int pa = pr == 0 ? 2 : (pr == 1 || pr >= 4 ? 0 : 1), pb = pr == 1 ? 2 : (pr == 2 || pr == 4 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < IB; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)      // operands swapped: a lane then holds 4 consecutive columns of one row
            acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][pb], a[i][pa], acc[ih + i][j], 0, 0, 0);
      }
    }
  };

  const int nfull = (a_in && b_in) ? ke / BK : 0;
  auto load_any = [&](int kt, float4 (&ra)[NA], float4 (&rb)[NB]) {
    if (kt >= nk) return;
    if (kt < nfull) load_tile(kt, std::true_type{}, ra, rb); else load_tile(kt, std::false_type{}, ra, rb);
  };
  load_any(0, ra0, rb0);
  split_store(ra0, rb0);
  __syncthreads();
  for (int t = 0; t < nk; t++) {
    load_any(t + 1, ra0, rb0);
    compute_tile();
    __syncthreads();
    if (t + 1 < nk) { split_store(ra0, rb0); __syncthreads(); }
  }

  // lane (c, q) holds C[wm0 + 16 i + c][wn0 + 16 j + 4 q + {0..3}]
  float* C = g.C;
  const bool vec = (g.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0 && (!g.mask || ((g.ldmask & 3) == 0 && (((uintptr_t)g.mask) & 15) == 0));
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int m = m0 + wm0 + 16 * i + c;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + wn0 + 16 * j + 4 * q;
      if (n >= g.N) continue;
      float* cp = C + (int64_t)m * g.ldc + n;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool whole = vec && n + 3 < g.N;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (n + e >= g.N) continue;
        if (g.epi == EPI_STORE) v[e] = act_apply(v[e] + (g.bias ? g.bias[n + e] : 0.0f), g.act);
        if (g.mask && !(g.mask[(int64_t)m * g.ldmask + n + e] > 0.0f)) v[e] = 0.0f;
        if (g.epi == EPI_ADD) v[e] = v[e] + cp[e];
        if (!whole) cp[e] = v[e];
      }
      if (whole) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// bf16-operand form of launch_gemm (tensor-op math mode): 128 x 128 tiles; EPI_ATOMIC splits K over workgroups
template <bool AKC, bool BKC, bool MASK_A = false>
int launch_gemm_bf16(ffh_ctx* c, GemmArgs& g, ffh_stream s, const char* name) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return FFH_OK;
  const bool x3 = ffh_split_mode(c);      // three bf16 planes per operand: fp32-accurate
  // 256 x 256 tiles (one 8-wave workgroup per CU, 128 x 64 per wave) where the output still fills the chip with them: half
  // the operand traffic per flop.  Measured at batch 32768: forward 3456 -> 1024 492 -> 353 us, 1024 -> 1024 162 -> 143;
  // dX +3...6 % from a reduction depth of 1024 up, slower below; the split-K weight-gradient form: see below.
  static const int tile_env = FFH_LAB_INT("FFH_BF16_TILE", 0);
  const int64_t tiles_big = (int64_t)((g.N + 255) / 256) * ((g.M + 255) / 256);
  const bool big_form = (AKC && BKC) || (AKC && !BKC && g.K >= 1024);
  // the split mode takes the same tile (96 KB: three planes per operand): forward / dX +4 %, the weight gradient of the big
  // layer 1,676 -> 1,241 us; whole split-mode step 6.9 -> 6.6-6.8 ms (FFH_X3_BIG: 0 never, 1 not for the weight gradient)
  static const int x3_big = FFH_LAB_INT("FFH_X3_BIG", 2);             // A/B switch
  bool big = (!x3 || x3_big) && big_form && g.M >= 256 && g.N >= 256 && tiles_big >= c->num_cus;
  // the split-K weight-gradient form: big tiles where there are enough of them that a split can fill whole rounds of one
  // workgroup per CU (3456 x 1024 at batch 32768: 56 tiles x 9 splits = 504; 747 -> 553 us); 16 tiles (1024 x 1024) lose 8 %
  static const int dw_big = FFH_LAB_INT("FFH_BF16_DW_BIG", 1);     // A/B switch
  if (dw_big && (!x3 || x3_big > 1) && !AKC && !BKC && g.epi == EPI_ATOMIC && !c->deterministic && tiles_big >= 32 && g.K >= 8192) big = true;
  if (tile_env == 128) big = false;
  if (tile_env == 256 && g.M >= 256 && g.N >= 256) big = true;
  const int bm = big ? 256 : kBfBM, bn = big ? 256 : kBfBN;
  const int gx = (g.N + bn - 1) / bn, gy = (g.M + bm - 1) / bm;
  g.splitk = 1;
  g.k_per_split = g.K;
  int gz = 1;
  if (g.epi == EPI_ATOMIC && c->deterministic) g.epi = EPI_ADD;
  if (g.epi == EPI_ATOMIC) {
    const int64_t tiles = (int64_t)gx * gy;
    int want = (int)(((big ? 1LL : 2LL) * c->num_cus + tiles - 1) / tiles);
    const int max_split = (g.K + 4 * kBfBK - 1) / (4 * kBfBK);     // at least four k-tiles per workgroup
    // (for the 128 x 128 tile the same search makes a weight-gradient GEMM up to 30 % faster ALONE and the step no faster:
    //  in the step the holes of an unbalanced launch are filled by the data-gradient GEMM on the other stream -- off by default)
    static const int fill_small = FFH_LAB_INT("FFH_BF16_FILL", 0);    // A/B switch
    if (big || fill_small) {
      // pick the split whose workgroup count fills whole rounds of resident workgroups best (one per CU for the big tile, two for 128 x 128)
      const int64_t slots = (big ? 1LL : 2LL) * c->num_cus;
      int best = want; double best_u = 0.0;
      for (int w2 = want; w2 <= 2 * want + 2 && w2 <= max_split; w2++) {
        const int64_t nb = tiles * w2;
        const double u = (double)nb / (double)(((nb + slots - 1) / slots) * slots);
        if (u > best_u + 1e-9) { best_u = u; best = w2; }
      }
      want = best;
    }
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int kps = (g.K + want - 1) / want;
    kps = (kps + kBfBK - 1) / kBfBK * kBfBK;
    g.k_per_split = kps;
    g.splitk = (g.K + kps - 1) / kps;
    if (g.splitk <= 1) g.splitk = 2;             // keeps blockIdx.z meaning "split" (the second split is empty)
    gz = g.splitk;
  }
  if (gy > 65535 || gz > 65535) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "gemm (bf16): grid too large");
  // split mode: the image of C (ffh_ctx_bf16x3_mirror_set) after a launch of the split-in-kernel form, by a pass over C
  auto x3_image_of_c = [&]() -> int {
    if (g.epi == EPI_ATOMIC) return FFH_OK;
    int col0 = 0;
    if (!ffh_planes_of(c, g.C, (size_t)((int64_t)(g.M - 1) * g.ldc + g.N) * 4, &col0)) return FFH_OK;
    return ffh_convert_f32_to_bf16x3(c, g.C, g.M, g.N, g.ldc, s);
  };
  if constexpr (!MASK_A) {
    if (x3) {      // both operands with images: the LDS-DMA form (linear_x3_dma.hip)
      const int rc = launch_gemm_x3_dma(c, g, AKC && BKC ? BF16_FORM_FWD : (AKC ? BF16_FORM_DX : BF16_FORM_DW), s, name);
      if (rc < 0) return rc;
      if (rc > 0) { if (g.db && !AKC && !BKC) g.db_done = 1; return FFH_OK; }
    }
  }
  if (x3) { char tok[96]; snprintf(tok, sizeof tok, "%s|bf16x3_%s|splitk=%d", name, (big && g.epi == EPI_ATOMIC) ? "256x256" : "128x128", g.splitk); ffh_route_add(c, tok); }
  if constexpr (AKC) {
    if (x3 && g.epi != EPI_ATOMIC) {      // forward / data gradient: the arithmetic of the LDS-DMA form, bit for bit (gemm_bf16x3_v2_kernel)
      // (128 x 128 tiles only: with 128 x 64 per wave the 16 x 16 accumulators, two operands' fragments and the staging registers do not fit 256 VGPRs)
      const int vx = (g.N + kBfBN - 1) / kBfBN, vy = (g.M + kBfBM - 1) / kBfBM;
      if (vy > 65535) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "gemm (bf16x3): grid too large");
      // few tiles (at most ~one per CU: 4096 x 3456 -> 1024 forward has 256): the same tile on EIGHT waves of 32 x 64 instead of four of 64 x 64 --
      // a CU that holds one workgroup then has two waves per SIMD, one's loads and LDS passes under the other's MFMAs (same sums per
      // accumulator, same bits)
      static const int w8_pct = FFH_LAB_INT("FFH_X3V_W8_PCT", 150);      // A/B switch: eight waves up to this many tiles, in per cent of the CUs
      if ((int64_t)vx * vy * 100 <= (int64_t)c->num_cus * w8_pct) {
        auto kv8 = gemm_bf16x3_v2_kernel<BKC, MASK_A, 128, 128, 32>;
        static const bool okv8 = glds_set_lds(kv8, kX3Lds);
        if (!okv8) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16x3): cannot reserve 48 KB of LDS");
        hipLaunchKernelGGL(kv8, dim3(vx, vy, 1), dim3(512), kX3Lds, as_stream(s), g);
      } else {
      auto kv = gemm_bf16x3_v2_kernel<BKC, MASK_A>;
      static const bool okv = glds_set_lds(kv, kX3Lds);
      if (!okv) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16x3): cannot reserve 48 KB of LDS");
      hipLaunchKernelGGL(kv, dim3(vx, vy, 1), dim3(256), kX3Lds, as_stream(s), g);
      }
      hipError_t ev = hipGetLastError();
      if (ev != hipSuccess) return ffh_fail_hip(c, ev, name);
      return x3_image_of_c();
    }
  }
  if (x3 && big) {
    auto kern3b = gemm_bf16x3_kernel<AKC, BKC, MASK_A, 256, 256, 128>;
    constexpr int lds3b = 3 * (256 + 256) * kX3BK * 2;          // 96 KB: one 8-wave workgroup per CU
    static const bool ok3b = glds_set_lds(kern3b, lds3b);
    if (!ok3b) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16x3): cannot reserve 96 KB of LDS");
    hipLaunchKernelGGL(kern3b, dim3(gx, gy, gz), dim3(512), lds3b, as_stream(s), g);
    hipError_t e3b = hipGetLastError();
    if (e3b != hipSuccess) return ffh_fail_hip(c, e3b, name);
    return x3_image_of_c();
  }
  if (x3) {
    auto kern3 = gemm_bf16x3_kernel<AKC, BKC, MASK_A>;
    static const bool ok3 = glds_set_lds(kern3, kX3Lds);
    if (!ok3) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16x3): cannot reserve 48 KB of LDS");
    hipLaunchKernelGGL(kern3, dim3(gx, gy, gz), dim3(256), kX3Lds, as_stream(s), g);
    hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) return ffh_fail_hip(c, e3, name);
    return x3_image_of_c();
  }
  // bf16 twins (ffh_ctx_bf16_mirror_set): the output's twin is written whenever one is registered; the operands come from
  // their twins when both have one and the problem is whole tiles of this launch
  g.A16 = g.B16 = nullptr; g.C16 = nullptr;
  if (!x3) {
    const bool akc = AKC, bkc = BKC;
    const int64_t lda = akc ? g.sAm : g.sAk, ldb = bkc ? g.sBn : g.sBk;
    const size_t a_span = (size_t)(((akc ? g.M : g.K) - 1) * lda + (akc ? g.K : g.M)) * 4, b_span = (size_t)(((bkc ? g.N : g.K) - 1) * ldb + (bkc ? g.K : g.N)) * 4;
    if (g.epi != EPI_ATOMIC) g.C16 = ffh_mirror_of(c, g.C, (size_t)((int64_t)(g.M - 1) * g.ldc + g.N) * 4);
    static const int no_src16 = FFH_LAB_INT("FFH_BF16_NO_TWINS", 0);      // A/B switch
    if (!MASK_A && !no_src16 && !g.a_not_twinned && g.M % 8 == 0 && g.N % 8 == 0 && g.M >= 8 && g.N >= 8 && g.K % kBfBK == 0 && g.k_per_split % kBfBK == 0 && lda % 8 == 0 && ldb % 8 == 0) {
      const unsigned short* a16 = ffh_mirror_of(c, g.A, a_span);
      const unsigned short* b16 = ffh_mirror_of(c, g.B, b_span);
      if (a16 && b16 && (((uintptr_t)a16 | (uintptr_t)b16) & 15) == 0) { g.A16 = a16; g.B16 = b16; }
    }
  }
  if constexpr (!MASK_A) {
    if (g.A16) {      // big outputs: the LDS-DMA kernel (linear_bf16_dma.hip)
      const int rc = launch_gemm_bf16_dma(c, g, AKC && BKC ? BF16_FORM_FWD : (AKC ? BF16_FORM_DX : BF16_FORM_DW), s, name);
      if (rc < 0) return rc;
      if (rc > 0) { if (g.db && !AKC && !BKC) g.db_done = 1; return FFH_OK; }
    }
    if (g.A16 && big) {
      auto kernw = gemm_bf16_kernel<AKC, BKC, false, 256, 256, 128, true>;
      static const bool okw = glds_set_lds(kernw, bf_lds_bytes(256, 256));
      if (!okw) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16): cannot reserve 128 KB of LDS");
      hipLaunchKernelGGL(kernw, dim3(gx, gy, gz), dim3(512), bf_lds_bytes(256, 256), as_stream(s), g);
      hipError_t ew = hipGetLastError();
      if (ew != hipSuccess) return ffh_fail_hip(c, ew, name);
      { char tok[96]; snprintf(tok, sizeof tok, "%s|bf16_256x256_twins|splitk=%d", name, g.splitk); ffh_route_add(c, tok); }
      return FFH_OK;
    }
    if (g.A16) {
      auto kern = gemm_bf16_kernel<AKC, BKC, false, kBfBM, kBfBN, 64, true>;
      static const bool ok = glds_set_lds(kern, kBfLds);
      if (!ok) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16): cannot reserve 64 KB of LDS");
      hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), kBfLds, as_stream(s), g);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return ffh_fail_hip(c, e, name);
      { char tok[96]; snprintf(tok, sizeof tok, "%s|bf16_128x128_twins|splitk=%d", name, g.splitk); ffh_route_add(c, tok); }
      return FFH_OK;
    }
  }
  { char tok[96]; snprintf(tok, sizeof tok, "%s|bf16%s_%s|splitk=%d", name, x3 ? "x3" : "", big ? "256x256" : "128x128", g.splitk); ffh_route_add(c, tok); }
  if (big && !x3) {
    auto kernw = gemm_bf16_kernel<AKC, BKC, MASK_A, 256, 256, 128>;
    static const bool okw = glds_set_lds(kernw, bf_lds_bytes(256, 256));
    if (!okw) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16): cannot reserve 128 KB of LDS");
    hipLaunchKernelGGL(kernw, dim3(gx, gy, gz), dim3(512), bf_lds_bytes(256, 256), as_stream(s), g);
    hipError_t ew = hipGetLastError();
    if (ew != hipSuccess) return ffh_fail_hip(c, ew, name);
    return FFH_OK;
  }
  auto kern = gemm_bf16_kernel<AKC, BKC, MASK_A>;
  static const bool ok = glds_set_lds(kern, kBfLds);
  if (!ok) return ffh_fail(c, FFH_ERR_HIP, "gemm (bf16): cannot reserve 64 KB of LDS");
  hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), kBfLds, as_stream(s), g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  return FFH_OK;
}

}  // namespace

namespace ffh_gemm {
int launch_gemm_bf16_form(ffh_ctx* c, GemmArgs& g, int form, ffh_stream s, const char* name) {
  switch (form) {
    case BF16_FORM_FWD: return launch_gemm_bf16<true, true>(c, g, s, name);
    case BF16_FORM_DW: return launch_gemm_bf16<false, false>(c, g, s, name);
    case BF16_FORM_DX: return launch_gemm_bf16<true, false>(c, g, s, name);
    case BF16_FORM_DX_MASK: return launch_gemm_bf16<true, false, true>(c, g, s, name);
  }
  return ffh_fail(c, FFH_ERR_BAD_ARG, "gemm (bf16): unknown form");
}
}  // namespace ffh_gemm
