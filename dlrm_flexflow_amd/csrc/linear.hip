// linear.hip -- Linear forward/backward and BatchMatmul on the gfx950 matrix cores, in
// exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; a k-ordered fmaf chain,
// bit-for-bit -- no xf32/TF32 exists on gfx950), 157 TFLOP/s dense peak.
//
// One LDS-tiled kernel serves every GEMM of the path through element strides:
//   C[m][n] (op)= sum_k A(m,k) * B(n,k),   A(m,k) = A[m*sAm + k*sAk],  B(n,k) = B[n*sBn + k*sBk]
//     Linear fwd  y  = x w^T        A = x  (k-contiguous)  B = w  (k-contiguous)   + bias + activation
//     Linear dx  += dy w            A = dy (k-contiguous)  B = w  (n-contiguous)
//     Linear dw  += dy^T x          A = dy (m-contiguous)  B = x  (n-contiguous)   split-K over the batch
//     BatchMatmul fwd/bwd           the same three forms with a batch stride (grid.z)
// Tiles are staged k-major in LDS ([k][m], [k][n]) so that the MFMA operand fetch is one
// conflict-free ds_read_b32 per lane whatever the global layout; a k-contiguous global tile is
// transposed while it is written to LDS (row pad 1 -> conflict-free scalar stores), an
// m/n-contiguous one is copied with 16-B stores (row pad 4).  Double-buffered LDS, global loads
// of tile t+1 issued before the MFMAs of tile t, one barrier per k-tile.
//
// Replaces cublasSgemm x2 + cudnnActivationForward [ref: src/ops/linear.cu:436-453],
// reluBackward/sigmoid_backward + cublasSgemm x2 + cublasSgemv [ref: src/ops/linear.cu:624-659],
// cublasSgemmStridedBatched [ref: src/ops/batch_matmul.cu:238-241,393-398].
#include "linear_gemm.h"

#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: a launch that carries its own completion event

#include <stdlib.h>
#include <type_traits>

using namespace ffh_gemm;

// The plain forward instantiation of the 128 x 128 x 16 kernel is compiled for four workgroups per CU (128 registers, 8 bytes of
// scratch) instead of the three its natural 164 registers allow: +1 % on the big layers' forward (1,737 -> 1,720 us), +0.35 %
// on the Terabyte step, three interleaved pairs.  The masking / weight-gradient forms lose as much under the same limit and
// keep theirs.  (Fewer than three workgroups per CU costs 6 % per workgroup: DESIGN 3.3.)
#ifndef FFH_FWD_OCC4
#define FFH_FWD_OCC4 1
#endif

namespace {


// SPLITW = false: 2x2 waves tile the BM x BN block, each wave owns (BM/2) x (BN/2) and the whole K.
// SPLITW = true : BM = BN = 32; the four waves share ONE 32x32 tile and split every k-tile four
//                 ways (intra-workgroup split-K), partial accumulators meet in LDS in a fixed order.
//                 For the skinny DLRM layers (2048 x 256, 2048 x 64 ...) this gives 4x the waves of
//                 a 64x64 tiling: a 32x32x2 MFMA chain over K = 512 alone is 16k cycles.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool SPLITW, bool FUSE_DY, bool CMAP>
__device__ __forceinline__ void gemm_f32_tile(const GemmArgs& g, const unsigned lin, const unsigned nbx, const unsigned nby, const unsigned nbz);

// One workgroup per tile (3-D grid), or -- g.tnx != 0 -- a persistent 1-D grid whose workgroups walk the tile space with
// stride gridDim.x: the hardware dispatcher hands out tiles of a multi-round launch unevenly over the CUs (a launch of 2048
// equal tiles at three resident workgroups per CU ends with some CUs a whole tile behind); a grid of exactly
// (resident workgroups per CU) x (CUs) workgroups, each with the same number of tiles, does not.  gridDim.x is a multiple of 8
// in that mode, so a workgroup's tiles stay in its XCD's contiguous range of the tile space.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool SPLITW = false, bool FUSE_DY = false, bool CMAP = false>
__global__ __launch_bounds__(256, (FFH_FWD_OCC4 && BM == 128 && BN == 128 && BK == 16 && AKC && BKC && !SPLITW && !FUSE_DY && !CMAP) ? 4 : 1)
void gemm_f32_kernel(const GemmArgs g) {
  ffh_kernel_prio();
  const bool pers = g.tnx != 0;
  const unsigned nbx = pers ? g.tnx : gridDim.x, nby = pers ? g.tny : gridDim.y, nbz = pers ? g.tnz : gridDim.z;
  const unsigned total = nbx * nby * nbz;
  const unsigned stride = pers ? gridDim.x : total;
  for (unsigned lin = pers ? blockIdx.x : (blockIdx.z * nby + blockIdx.y) * nbx + blockIdx.x; lin < total; lin += stride) {
    gemm_f32_tile<BM, BN, BK, AKC, BKC, SPLITW, FUSE_DY, CMAP>(g, lin, nbx, nby, nbz);
    if (lin + stride < total) __syncthreads();   // the epilogue's LDS exchanges (bias sums, split-wave reduction) end before the next tile stages
  }
}

template <int BM, int BN, int BK, bool AKC, bool BKC, bool SPLITW, bool FUSE_DY, bool CMAP>
__device__ __forceinline__ void gemm_f32_tile(const GemmArgs& g, const unsigned lin, const unsigned nbx, const unsigned nby, const unsigned nbz) {
  constexpr int PA = AKC ? 1 : 4, PB = BKC ? 1 : 4;
  constexpr int LA = BM + PA, LB = BN + PB;
  constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;   // float4 staging slots per thread
  constexpr int KQ = BK / 4;                                // float4 per k-contiguous row
  constexpr int RPP = 256 / KQ;                             // rows per staging pass
  constexpr int WM = SPLITW ? 1 : BM / 64, WN = SPLITW ? 1 : BN / 64;   // 32x32 MFMA tiles per wave
  static_assert(!SPLITW || (BM == 32 && BN == 32 && BK % 8 == 0), "split-wave form is 32x32");
  __shared__ float As[2][BK * LA];
  __shared__ float Bs[2][BK * LB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs in
  // dispatch order, so XCD x sees linear ids x, x+8, ...  Give each XCD one CONTIGUOUS range of
  // the (z, y, x) tile space instead: its A rows / its K-split are then private to its L2 and only
  // the small shared operand is fetched by all eight.
  int bx, by, bz;
  {
    const unsigned total = nbx * nby * nbz;
    const unsigned xcd = lin & 7u, loc = lin >> 3;
    const unsigned q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx);
    by = (int)((nlin / nbx) % nby);
    bz = (int)(nlin / (nbx * nby));
  }
  const int m0 = by * BM, n0 = bx * BN;
  const float* A = g.A;
  const float* B = g.B;
  float* C = g.C;
  int kb = 0, ke = g.K;
  if (g.splitk > 1) {
    kb = bz * g.k_per_split;
    ke = kb + g.k_per_split < g.K ? kb + g.k_per_split : g.K;
  } else {
    A += (int64_t)bz * g.bsA; B += (int64_t)bz * g.bsB; C += (int64_t)bz * g.bsC;
  }
  if (kb >= ke) return;
  const int nk = (ke - kb + BK - 1) / BK;

  // 16-byte global accesses need only 4-byte alignment on gfx950 (ld4u / st4u): rows of 857 or 13 floats take the same
  // unguarded dwordx4 loads as aligned ones
  constexpr bool a_vec = true, b_vec = true;

  float4 ra2[2][NA], rb2[2][NB];     // two tiles in flight in registers (set = stage parity)
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr bool y_vec = FUSE_DY;

  // interior tiles (the common case) take unguarded 16-B loads: the branch is workgroup-uniform
  const bool a_in = a_vec && (m0 + BM <= g.M);
  const bool b_in = b_vec && (n0 + BN <= g.N);

  // k-tiles that lie completely inside the matrices take the unguarded body (FAST): no per-element
  // predicates, plain global_load_dwordx4.  The choice is workgroup-uniform.
  const int nfull = (a_in && b_in && (!FUSE_DY || y_vec)) ? (ke - kb) / BK : 0;

  auto load_tile_t = [&](int kt, auto fast_tag, auto set_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    float4 (&ra)[NA] = ra2[decltype(set_tag)::value];
    float4 (&rb)[NB] = rb2[decltype(set_tag)::value];
    const int k0 = kb + kt * BK;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {   // rows of A are k-contiguous: BK/4 float4 per row
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        const int m = m0 + row, k = k0 + k4 * 4;
        if (FAST) ra[i] = ld4u(A + (int64_t)m * g.sAm + k);
        else ra[i] = load4_guard(A + (int64_t)m * g.sAm + k, m < g.M, k, ke, a_vec);
        if (FUSE_DY && (g.fuse & 1)) {
          // dx form: relu' applied to dy as it is loaded (the in-place result belongs to the dw GEMM,
          // which may run concurrently on another stream)
          float4 yv;
          if (FAST) yv = ld4u(g.act_y + (int64_t)m * g.ld_act_y + k);
          else yv = load4_guard(g.act_y + (int64_t)m * g.ld_act_y + k, m < g.M, k, ke, y_vec);
          ra[i].x = yv.x > 0.0f ? ra[i].x : 0.0f; ra[i].y = yv.y > 0.0f ? ra[i].y : 0.0f;
          ra[i].z = yv.z > 0.0f ? ra[i].z : 0.0f; ra[i].w = yv.w > 0.0f ? ra[i].w : 0.0f;
        }
      } else {     // rows of the tile are k, contiguous along m
        constexpr int PER = BM / 4;
        const int m4 = tid % PER, kr = tid / PER + i * (256 / PER);
        const int m = m0 + m4 * 4, k = k0 + kr;
        if (FAST) ra[i] = ld4u(A + (int64_t)k * g.sAk + m);
        else ra[i] = load4_guard(A + (int64_t)k * g.sAk + m, k < ke, m, g.M, a_vec);
        if (FUSE_DY && !AKC) {
          if (g.fuse & 1) {
            // reluBackward [ref: src/runtime/cuda_helper.cu:71-78] on the fly; idempotent, so the
            // in-place write-back by column 0 may race with the other columns' reads harmlessly
            float4 yv;
            if (FAST) yv = ld4u(g.act_y + (int64_t)k * g.ld_act_y + m);
            else yv = load4_guard(g.act_y + (int64_t)k * g.ld_act_y + m, k < ke, m, g.M, y_vec);
            ra[i].x = yv.x > 0.0f ? ra[i].x : 0.0f; ra[i].y = yv.y > 0.0f ? ra[i].y : 0.0f;
            ra[i].z = yv.z > 0.0f ? ra[i].z : 0.0f; ra[i].w = yv.w > 0.0f ? ra[i].w : 0.0f;
            if (bx == 0 && (FAST || k < ke)) {
              float* wp = const_cast<float*>(A) + (int64_t)k * g.sAk + m;
              if (FAST || (a_vec && m + 3 < g.M)) st4u(wp, ra[i]);
              else {
                if (m + 0 < g.M) wp[0] = ra[i].x;
                if (m + 1 < g.M) wp[1] = ra[i].y;
                if (m + 2 < g.M) wp[2] = ra[i].z;
                if (m + 3 < g.M) wp[3] = ra[i].w;
              }
            }
          }
          if ((g.fuse & 2) && bx == 0) { bsum.x += ra[i].x; bsum.y += ra[i].y; bsum.z += ra[i].z; bsum.w += ra[i].w; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        const int n = n0 + row, k = k0 + k4 * 4;
        if (FAST) rb[i] = ld4u(B + (int64_t)n * g.sBn + k);
        else rb[i] = load4_guard(B + (int64_t)n * g.sBn + k, n < g.N, k, ke, b_vec);
      } else {
        constexpr int PER = BN / 4;
        const int n4 = tid % PER, kr = tid / PER + i * (256 / PER);
        const int n = n0 + n4 * 4, k = k0 + kr;
        if (FAST) rb[i] = ld4u(B + (int64_t)k * g.sBk + n);
        else rb[i] = load4_guard(B + (int64_t)k * g.sBk + n, k < ke, n, g.N, b_vec);
      }
    }
  };

  auto store_tile = [&](int buf, auto set_tag) {
    float4 (&ra)[NA] = ra2[decltype(set_tag)::value];
    float4 (&rb)[NB] = rb2[decltype(set_tag)::value];
    float* as = As[buf];
    float* bs = Bs[buf];
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        as[(k4 * 4 + 0) * LA + row] = ra[i].x;
        as[(k4 * 4 + 1) * LA + row] = ra[i].y;
        as[(k4 * 4 + 2) * LA + row] = ra[i].z;
        as[(k4 * 4 + 3) * LA + row] = ra[i].w;
      } else {
        constexpr int PER = BM / 4;
        const int m4 = tid % PER, kr = tid / PER + i * (256 / PER);
        *reinterpret_cast<float4*>(&as[kr * LA + m4 * 4]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        bs[(k4 * 4 + 0) * LB + row] = rb[i].x;
        bs[(k4 * 4 + 1) * LB + row] = rb[i].y;
        bs[(k4 * 4 + 2) * LB + row] = rb[i].z;
        bs[(k4 * 4 + 3) * LB + row] = rb[i].w;
      } else {
        constexpr int PER = BN / 4;
        const int n4 = tid % PER, kr = tid / PER + i * (256 / PER);
        *reinterpret_cast<float4*>(&bs[kr * LB + n4 * 4]) = rb[i];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  const int wm0 = SPLITW ? 0 : (wave >> 1) * (BM / 2), wn0 = SPLITW ? 0 : (wave & 1) * (BN / 2);
  constexpr int KPAIRS = SPLITW ? BK / 8 : BK / 2;     // k-pairs (MFMA steps) per wave per k-tile
  const int kk0 = SPLITW ? wave * KPAIRS : 0;
  const int lr = lane & 31, lh = lane >> 5;

  auto compute_tile = [&](int buf) {
    const float* as = As[buf];
    const float* bs = Bs[buf];
    // all operand fetches of the k-tile first (counted lgkmcnt waits), then the MFMA chain
    constexpr int KH = KPAIRS > 8 ? 8 : KPAIRS;     // fetch in groups of <= 8 k-pairs to bound registers
#pragma unroll
    for (int kg = 0; kg < KPAIRS; kg += KH) {
      float a[KH][WM], b[KH][WN];
#pragma unroll
      for (int kq = 0; kq < KH; kq++) {
        const int kk = kk0 + kg + kq;
#pragma unroll
        for (int i = 0; i < WM; i++) a[kq][i] = as[(2 * kk + lh) * LA + wm0 + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < WN; j++) b[kq][j] = bs[(2 * kk + lh) * LB + wn0 + j * 32 + lr];
      }
#pragma unroll
      for (int kq = 0; kq < KH; kq++)
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
          for (int j = 0; j < WN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kq][i], b[kq][j], acc[i][j], 0, 0, 0);
    }
  };

  // phase 1: software pipeline over "stages": the interior k-tiles, preceded by the single K-tail tile when
  // there is exactly one (K = 432: 6 full tiles + 48).  TWO stages are in flight in registers while the
  // MFMAs run on a third in LDS: one k-tile of MFMA work (0.2-0.4 us) is far shorter than an L2/HBM round
  // trip under load, so a single prefetched tile leaves the matrix pipe waiting.  Only stage 0 may take the
  // guarded loader; the loop body has unguarded loads only, so no control-flow merge drains vmcnt early.
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int tail_first = (nfull > 0 && nk == nfull + 1) ? 1 : 0;
  const int nstages = nfull > 0 ? nfull + tail_first : 0;
  if (nstages > 0) {
    if (tail_first) load_tile_t(nk - 1, std::false_type{}, S0{});
    else load_tile_t(0, std::true_type{}, S0{});
    if (nstages > 1) load_tile_t(1 - tail_first, std::true_type{}, S1{});
    store_tile(0, S0{});
    __syncthreads();
    int i = 0;
    // steady state, branch-free: both prefetches exist (straight-line code keeps the accumulators in AGPRs and
    // lets the scheduler place the LDS stores of stage i+1 behind the MFMAs of stage i)
    for (; i + 3 < nstages; i += 2) {
      load_tile_t(i + 2 - tail_first, std::true_type{}, S0{});
      compute_tile(0);
      store_tile(1, S1{});
      __syncthreads();
      load_tile_t(i + 3 - tail_first, std::true_type{}, S1{});
      compute_tile(1);
      store_tile(0, S0{});
      __syncthreads();
    }
    // drain: the last (up to three) stages
    for (; i < nstages; i += 2) {
      // even stage: LDS buffer 0 holds stage i, register set 1 holds stage i+1
      if (i + 2 < nstages) load_tile_t(i + 2 - tail_first, std::true_type{}, S0{});
      compute_tile(0);
      if (i + 1 < nstages) store_tile(1, S1{});
      __syncthreads();
      if (i + 1 >= nstages) break;
      // odd stage: LDS buffer 1 holds stage i+1, register set 0 holds stage i+2
      if (i + 3 < nstages) load_tile_t(i + 3 - tail_first, std::true_type{}, S1{});
      compute_tile(1);
      if (i + 2 < nstages) store_tile(0, S0{});
      __syncthreads();
    }
  }
  // phase 2: the remaining edge tiles (ragged M/N, unaligned operands, more than one partial k-tile): guarded, not pipelined
  for (int kt = nfull; kt < nk - tail_first; kt++) {
    load_tile_t(kt, std::false_type{}, S0{});
    store_tile(0, S0{});
    __syncthreads();
    compute_tile(0);
    __syncthreads();
  }

  if (FUSE_DY && !AKC) {
    if ((g.fuse & 2) && bx == 0) {
      // bias gradient: the threads that staged the same 4 columns (different k rows) meet in LDS
      constexpr int PER = BM / 4;
      float4* sb = reinterpret_cast<float4*>(As[0]);
      sb[tid] = bsum;
      __syncthreads();
      if (tid < PER) {
        float4 t = sb[tid];
        for (int q = 1; q < 256 / PER; q++) { const float4 u = sb[q * PER + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const int m = m0 + tid * 4;
        if (m + 0 < g.M) atomicAdd(&g.db[m + 0], t.x);
        if (m + 1 < g.M) atomicAdd(&g.db[m + 1], t.y);
        if (m + 2 < g.M) atomicAdd(&g.db[m + 2], t.z);
        if (m + 3 < g.M) atomicAdd(&g.db[m + 3], t.w);
      }
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
  if (SPLITW) {
    // the four k-slices of the tile: ((w0 + w1) + w2) + w3, each wave finishes 4 of the 16 registers.
    // The staging buffers are dead by now (the k loop ended on a barrier): reuse A's as the exchange area, so
    // the kernel stays at 34 KB of LDS = four workgroups per CU (1024 tiles of a 2048 x 512 layer in ONE round).
    static_assert(2 * BK * LA >= 4 * 16 * 64, "A staging area too small for the split-wave reduction");
    float* red = &As[0][0];
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    const int n = n0 + lr;
    if (n < g.N) {
      const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = wave * 4 + q;
        const float v = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) + red[(3 * 16 + r) * 64 + lane];
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* cp = C + (int64_t)m * g.ldc + n;
        const float vm = (g.mask && !(g.mask[(int64_t)m * g.ldmask + n] > 0.0f)) ? 0.0f : v;
        if (g.epi == EPI_STORE) *cp = act_apply(vm + bv, g.act);
        else if (g.epi == EPI_ADD) *cp = *cp + vm;
        else atomicAdd(cp, vm);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      if (n >= g.N) continue;
      const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
      float* cbase = C + n;
      int64_t cld = g.ldc;
      if (CMAP) { const ffh_col_dest cd = g.colmap[n]; cbase = cd.base; cld = cd.ld; }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* cp = cbase + (int64_t)m * cld;
        float v = acc[i][j][r];
        if (g.mask && !(g.mask[(int64_t)m * g.ldmask + n] > 0.0f)) v = 0.0f;
        if (g.epi == EPI_STORE) *cp = act_apply(v + bv, g.act);
        else if (g.epi == EPI_ADD) *cp = *cp + v;
        else atomicAdd(cp, v);
      }
    }
}

// =============================================================================================
// LDS-DMA GEMM for the mid-size layers of the 2048-sample step (432x512, 512x256 ...): outputs of only
// ~1 M elements, i.e. ONE 64x64 tile per CU.  What bounds such a launch is not the matrix pipe but how the
// operand bytes get on chip next to it, so:
//   * operands go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass), three
//     k-tiles of 64 in flight, counted s_waitcnt vmcnt + one raw s_barrier per k-tile;
//   * a workgroup is 16 waves on one tile: 4 (or 2) sub-tiles of 32x32 x 4 (or 8) k-slices of every k-tile.
//     Issuing one 1-KiB DMA piece costs its wave ~60-180 cycles; with four waves per SIMD that time sits
//     under the other waves' MFMAs (measured: 4 waves 16.0 us, 8 waves 14.1, 16 waves 12.9 on 2048x512x432);
//   * the DMA pieces of a k-tile are issued between the MFMAs of the previous one, not in a burst;
//   * LDS images are lane-linear (a DMA writes base + 16*lane); any layout freedom is taken on the SOURCE
//     address.  A k-contiguous operand (rows = m) is XOR-swizzled by 16-B chunk so that the MFMA operand fetch
//     is one conflict-free ds_read_b128 per FOUR k-steps -- legal because a sum over k may visit k in any
//     order as long as both operands agree: lane (row, half h) of the 32x32x2 MFMA takes k = 8j + 4h + e.
//     An m/n-contiguous operand (rows = k) is read with one ds_read_b32 per k-step in that same k order;
//   * rows / k beyond the matrix come from a 256-byte zero page (per-lane source address), so edges need no
//     predicated code; the k-slices meet in LDS in a fixed order; bias, activation, relu'(x) mask or atomics
//     in the epilogue; db (column sums of dy) is summed from the LDS image by the first column of workgroups.
// Results differ from the register-staged kernel above only in the order of the k sum (both are exact-fp32
// fmaf chains); parity tests compare both against the oracle at 1e-5.
// =============================================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct GldsArgs {
  const float* A;  const float* B;  float* C;
  const float* bias;
  const float* zeros;        // >= 16 readable zero bytes
  const float* mask;         // epilogue: C = mask[m][n] > 0 ? v : 0 (relu' of the layer below), or null
  float*       db;           // dW form: db[m] += sum_k A(m,k), or null
  int64_t lda, ldb, ldc, ldmask;
  int M, N, K;
  int k_per_split, splitk;   // grid.z = splitk; k_per_split is a multiple of 64
  int epi, act;
  const ffh_col_dest* colmap;   // epilogue: column n of C lives at colmap[n].base[m * colmap[n].ld] (a Concat backward folded in), or null
};

template <int N_> __device__ __forceinline__ void glds_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

constexpr int kGldsWaves = 16;
constexpr int kGldsStages = 3;

// AKR / BKR: false = rows of the operand are m (n), k contiguous (x, w in fwd; dy in dx); true = rows are k, m (n) contiguous.
// body of one workgroup: `lin` is its index in the nbx x nby x nbz tile space of THIS problem (a launch may carry two)
// NSTAGE: LDS stages of the k pipeline.  3 (97 KB) when a CU holds one workgroup anyway; 2 (65 KB) lets two workgroups share a
// CU, so that one's prologue / epilogue runs under the other's main loop (launches with more workgroups than CUs)
template <bool AKR, bool BKR, int BM, int NSTAGE = kGldsStages>
__device__ __forceinline__ void glds_body(const GldsArgs& g, const unsigned lin, const unsigned nbx, const unsigned nby, const unsigned nbz) {
  constexpr int NW = kGldsWaves, BN = 64;
  constexpr int NSUB = (BM / 32) * 2;                   // 32x32 sub-tiles of the block tile
  constexpr int KS = NW / NSUB;                         // k-slices of every k-tile
  constexpr int JW = 8 / KS;                            // k-octets per wave per k-tile
  constexpr int A_CH = BM * 16, B_CH = BN * 16, STAGE_CH = A_CH + B_CH;   // 16-byte chunks
  constexpr int NPIECE = STAGE_CH / 64;
  constexpr int NIW = (NPIECE + NW - 1) / NW;           // DMA pieces per k-tile per wave (the last may be absent)
  static_assert(JW >= 1 && 8 % KS == 0, "k-slices");
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by, bz;
  {
    const unsigned total = nbx * nby * nbz;
    if (lin >= total) return;                            // padding workgroup of a two-problem launch
    const unsigned xcd = lin & 7u, loc = lin >> 3, q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx); by = (int)((nlin / nbx) % nby); bz = (int)(nlin / (nbx * nby));
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kb = bz * g.k_per_split;
  const int ke = kb + g.k_per_split < g.K ? kb + g.k_per_split : g.K;
  if (kb >= ke) return;
  const int nk = (ke - kb + 63) / 64;
  const int sub = wave % NSUB, ks = wave / NSUB;
  const int wm = sub >> 1, wn = sub & 1;

  // per-lane source of this wave's DMA pieces.  Piece q holds chunks 64q .. 64q+63 of the stage image:
  //   rows-are-m image: chunk = row * 16 + slot, slot = kchunk ^ (row & 15)  (k-tile = 16 chunks of 4 k)
  //   rows-are-k image: chunk = krow * (cols/4) + mchunk                      (no swizzle: read by ds_read_b32 along m)
  const float* src[NIW];
  int src_k[NIW];          // k offset of the lane's chunk inside the k-tile
  bool src_on[NIW];
#pragma unroll
  for (int i = 0; i < NIW; i++) {
    const int q = wave + NW * i;
    src_on[i] = q < NPIECE;
    const int ch = q * 64 + lane;
    const bool isA = ch < A_CH;
    const int cb = isA ? ch : ch - A_CH;
    const float* base = isA ? g.A : g.B;
    const int64_t ld = isA ? g.lda : g.ldb;
    const int o0 = isA ? m0 : n0, olim = isA ? g.M : g.N;
    const bool kr = isA ? AKR : BKR;
    const int cols4 = (isA ? BM : BN) / 4;
    if (!kr) {
      const int r = cb >> 4, c = (cb & 15) ^ (r & 15);
      src[i] = (o0 + r < olim) ? base + (int64_t)(o0 + r) * ld + 4 * c : nullptr;
      src_k[i] = 4 * c;
    } else {
      const int kr_ = cb / cols4, c = cb - kr_ * cols4;
      src[i] = (o0 + 4 * c < olim) ? base + (int64_t)kr_ * ld + o0 + 4 * c : nullptr;
      src_k[i] = kr_ | (1 << 30);       // flag: the k offset moves the row, not the column
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
  auto issue = [&](int kt, int buf, int i0, int i1) {
    const int k0 = kb + kt * 64;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      const int q = wave + NW * i;
      const bool rows_k = (src_k[i] >> 30) & 1;
      const int kk = k0 + (src_k[i] & 0xFFFF);
      const int64_t ld = (q * 64 < A_CH) ? g.lda : g.ldb;
      const float* p = (src_on[i] && src[i] != nullptr && kk < ke) ? (rows_k ? src[i] + (int64_t)k0 * ld : src[i] + k0) : g.zeros;
      // a wave without an i-th piece (24 pieces on 16 waves) still issues one, into a spare KiB behind the stages: every
      // wave then has the same number of DMAs per k-tile and one counted vmcnt wait serves all of them
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(src_on[i] ? buf * STAGE_CH + q * 64 : NSTAGE * STAGE_CH) * 16u);
      unsigned keep;
      // inline asm on purpose: hipcc's waitcnt pass must not see the LDS-DMA, or it drains vmcnt(0) before every ds_read
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  float dbsum = 0.0f;
  const bool do_db = AKR && g.db != nullptr && bx == 0;
  constexpr int DB_GROUPS = NW * 64 / BM, DB_ROWS = 64 / DB_GROUPS;

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) issue(s, s, 0, NIW);
  const int ra = wm * 32 + lr, rb = wn * 32 + lr;
  int buf = 0, nbuf = NSTAGE - 1;
  for (int t = 0; t < nk; t++) {
    glds_wait_vmcnt<(NSTAGE - 2) * NIW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* as4 = lds + buf * STAGE_CH;
    const float4* bs4 = as4 + A_CH;
    const float* as1 = reinterpret_cast<const float*>(as4);
    const float* bs1 = reinterpret_cast<const float*>(bs4);
    auto rd = [&](int j, float4& a, float4& b) {
      const int oct = ks * JW + j;
      if (!AKR) a = as4[ra * 16 + ((2 * oct + lh) ^ (ra & 15))];
      else {
        const float* q = as1 + (8 * oct + 4 * lh) * BM + ra;
        a = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
      }
      if (!BKR) b = bs4[rb * 16 + ((2 * oct + lh) ^ (rb & 15))];
      else {
        const float* q = bs1 + (8 * oct + 4 * lh) * BN + rb;
        b = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
      }
    };
    float4 a_cur, b_cur, a_nxt, b_nxt;
    rd(0, a_cur, b_cur);
#pragma unroll
    for (int j = 0; j < JW; j++) {
      if (j + 1 < JW) rd(j + 1, a_nxt, b_nxt);
      issue(t + NSTAGE - 1, nbuf, j * NIW / JW, (j + 1) * NIW / JW);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc, 0, 0, 0);
      a_cur = a_nxt; b_cur = b_nxt;
    }
    if (do_db) {
      // bias gradient: column sums of the dy image (rows = k): thread -> column tid % BM, rows (tid / BM) * DB_ROWS ...
      const int col = tid % BM, r0 = (tid / BM) * DB_ROWS;
#pragma unroll
      for (int r = 0; r < DB_ROWS; r++) dbsum += as1[(r0 + r) * BM + col];
    }
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }
  glds_wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // the k-slices of a sub-tile meet in LDS, ((s0 + s1) + s2) + ..., each wave finishes 16 / KS accumulator registers
  constexpr int RPW = 16 / KS;
  float* red = reinterpret_cast<float*>(lds);        // [ks][sub][16][64] floats = NW * 4 KB
  float out[RPW];
  if (KS > 1) {
#pragma unroll
    for (int r = 0; r < 16; r++) red[((ks * NSUB + sub) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPW; q++) {
      const int r = ks * RPW + q;
      float v = red[((0 * NSUB + sub) * 16 + r) * 64 + lane];
#pragma unroll
      for (int s2 = 1; s2 < KS; s2++) v += red[((s2 * NSUB + sub) * 16 + r) * 64 + lane];
      out[q] = v;
    }
  } else {
#pragma unroll
    for (int q = 0; q < RPW; q++) out[q] = acc[q];
  }
  if (do_db) {
    __syncthreads();
    red[tid] = dbsum;                                  // [DB_GROUPS][BM]
    __syncthreads();
    if (tid < BM) {
      float v = red[tid];
      for (int q = 1; q < DB_GROUPS; q++) v += red[q * BM + tid];
      if (m0 + tid < g.M) atomicAdd(&g.db[m0 + tid], v);
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
  const int n = n0 + wn * 32 + lr;
  if (n < g.N) {
    const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
    float* cbase = g.C + n;
    int64_t cld = g.ldc;
    if (g.colmap) { const ffh_col_dest cd = g.colmap[n]; cbase = cd.base; cld = cd.ld; }
#pragma unroll
    for (int q = 0; q < RPW; q++) {
      const int r = KS > 1 ? ks * RPW + q : q;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= g.M) continue;
      float v = out[q];
      if (g.mask && !(g.mask[(int64_t)m * g.ldmask + n] > 0.0f)) v = 0.0f;
      float* cp = cbase + (int64_t)m * cld;
      if (g.epi == EPI_STORE) *cp = act_apply(v + bv, g.act);
      else if (g.epi == EPI_ADD) *cp = *cp + v;
      else atomicAdd(cp, v);
    }
  }
}

template <bool AKR, bool BKR, int BM, int NSTAGE = kGldsStages>
__global__ __launch_bounds__(kGldsWaves * 64) void gemm_glds_kernel(const GldsArgs g) {
  ffh_kernel_prio();
  glds_body<AKR, BKR, BM, NSTAGE>(g, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, gridDim.z);
}

// One launch for a layer's backward: workgroups [0, na8) are the data-gradient GEMM (dy x w, rows = samples), the rest the
// weight-gradient GEMM (dy^T x x, split-K over the batch).  Both read the same dy; no second stream, no fork / join
// events (each is a barrier packet on the critical stream), one dispatch instead of two.  na8 = the dX tile count
// rounded up to a multiple of 8 so that both problems keep their XCD-contiguous tile order.
struct GldsDims { unsigned nbx, nby, nbz; };
template <int BM_DX, int NSTAGE = kGldsStages>
__global__ __launch_bounds__(kGldsWaves * 64) void gemm_glds_bwd_kernel(const GldsArgs dxg, const GldsDims dxd, const unsigned na8,
                                                                        const GldsArgs dwg, const GldsDims dwd) {
  ffh_kernel_prio();
  if (blockIdx.x < na8) glds_body<false, true, BM_DX, NSTAGE>(dxg, blockIdx.x, dxd.nbx, dxd.nby, dxd.nbz);
  else glds_body<true, true, 64, NSTAGE>(dwg, blockIdx.x - na8, dwd.nbx, dwd.nby, dwd.nbz);
}

inline bool glds_aligned(const float* p, int64_t ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

struct GldsPlan { int bm; dim3 grid; int lds_bytes; };

// Fills the plan (tile height, grid, split-K) if the LDS-DMA kernel should take this GEMM.
template <bool AKR, bool BKR>
bool plan_glds(ffh_ctx* c, GldsArgs& g, bool atomic_splitk, GldsPlan& p, double min_work = 1.5e8, int min_k = 128) {
  static const int off = FFH_LAB_INT("FFH_GEMM_NO_GLDS", 0);   // A/B switch (tools/ab.sh)
  if (off || !c->zeros) return false;
  if (c->deterministic && atomic_splitk) return false;
  // weight-gradient GEMMs of the big-batch steps: above ~1e9 MACs the register-staged 128 x 128 split-K kernel wins (whole
  // Terabyte-shape step, interleaved A/B on one box: 1.27 vs 1.36 ms at 4096 samples, 2.48 vs 2.59 at 8192, 4.62 vs 4.78 at
  // 16384) -- except at 32768 samples, where this kernel's 16-wave workgroups share the chip better with the dX GEMM running
  // beside them on the other stream (9.11 vs 9.24-9.31 ms), so there it stays
  static const double dw_max = FFH_LAB_F64("FFH_GLDS_DW_MAX", 1.0e9);   // A/B switches
  static const int dw_kmax = FFH_LAB_INT("FFH_GLDS_DW_KMAX", 16384);
  if (atomic_splitk && (double)g.M * g.N * g.K >= dw_max && g.K <= dw_kmax) return false;
  if (!glds_aligned(g.A, g.lda) || !glds_aligned(g.B, g.ldb)) return false;
  // whole 16-byte chunks only: the contiguous extent of each operand must be a multiple of 4 floats
  if ((AKR ? g.M : g.K) % 4 || (BKR ? g.N : g.K) % 4) return false;
  const double work = (double)g.M * g.N * g.K;
  if (work < min_work || g.M < 64 || g.N < 64 || g.K < min_k) return false;   // the small layers are launch-bound either way
  const int64_t tiles64 = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64);
  int bm = 64;
  int64_t tiles = tiles64;
  if (!atomic_splitk && tiles64 < (3 * c->num_cus) / 4) { bm = 32; tiles = (int64_t)((g.M + 31) / 32) * ((g.N + 63) / 64); }
  // bigger GEMMs: the register-staged kernels with their larger tiles (measured again with the two-stage form at the
  // MLPerf / Terabyte layer sizes: 3-8 % slower per step through this kernel).  Splitting a 256-tile forward GEMM into
  // 512 half-height tiles to get two workgroups per CU also lost (Kaggle step +16 us).
  if (tiles > (3 * c->num_cus) / 2) return false;
  g.zeros = c->zeros;
  g.splitk = 1; g.k_per_split = (g.K + 63) / 64 * 64;
  if (atomic_splitk) {
    int want = (int)(c->num_cus / tiles);               // one round of workgroups: never more of them than CUs
    const int max_split = (g.K + 255) / 256;            // at least four k-tiles per workgroup
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int kps = (g.K + want - 1) / want;
    kps = (kps + 63) / 64 * 64;
    g.k_per_split = kps;
    g.splitk = (g.K + kps - 1) / kps;
  }
  const int gy = (g.M + bm - 1) / bm, gx = (g.N + 63) / 64;
  if (gy > 65535 || g.splitk > 65535) return false;
  p.bm = bm;
  p.grid = dim3(gx, gy, g.splitk);
  p.lds_bytes = kGldsStages * (bm + 64) * 256 + 1024;
  return true;
}


// Returns 1 if the LDS-DMA kernel took the GEMM, 0 if the shape is not its business, < 0 on error.
template <bool AKR, bool BKR>
int launch_glds(ffh_ctx* c, GldsArgs& g, bool atomic_splitk, ffh_stream s, const char* name) {
  GldsPlan p;
  if (!plan_glds<AKR, BKR>(c, g, atomic_splitk, p)) return 0;
  // more workgroups than CUs: two stages (65 KB) so that two of them share a CU
  const bool two = (int64_t)p.grid.x * p.grid.y * p.grid.z > (int64_t)c->num_cus;
  const int lds2 = 2 * 128 * 256 + 1024;       // the epilogue's cross-wave reduction needs 64 KB whatever the tile
#define FFH_GLDS_LAUNCH(BMV, NS, LDSV)                                                                                     \
  {                                                                                                                        \
    auto kern = gemm_glds_kernel<AKR, BKR, BMV, NS>;                                                                       \
    static const bool ok = glds_set_lds(kern, NS * 128 * 256 + 1024);                                                      \
    if (!ok) return 0;                                                                                                     \
    hipLaunchKernelGGL(kern, p.grid, dim3(kGldsWaves * 64), LDSV, as_stream(s), g);                                        \
  }
  if (p.bm == 64) { if (two) FFH_GLDS_LAUNCH(64, 2, lds2) else FFH_GLDS_LAUNCH(64, 3, p.lds_bytes) }
  else { if (two) FFH_GLDS_LAUNCH(32, 2, lds2) else FFH_GLDS_LAUNCH(32, 3, p.lds_bytes) }
#undef FFH_GLDS_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  { char tok[96]; snprintf(tok, sizeof tok, "%s|glds_%dx64_s%d|splitk=%d", name, p.bm, two ? 2 : 3, g.splitk); ffh_route_add(c, tok); }
  return 1;
}

// dX and dW of one layer in ONE launch (gemm_glds_bwd_kernel); 1 = done, 0 = not applicable
int launch_glds_bwd(ffh_ctx* c, GldsArgs& dxg, GldsArgs& dwg, ffh_stream s) {
  static const int off = FFH_LAB_INT("FFH_GLDS_NO_DUAL", 0);   // A/B switch (tools/ab.sh)
  if (off || c->deterministic) return 0;
  GldsPlan px, pw;
  // (the 256x64 layer was tried as a pair too: no gain, so the work threshold of the single GEMMs stands; k >= 64 suffices)
  if (!plan_glds<false, true>(c, dxg, false, px, 1.5e8, 64) || !plan_glds<true, true>(c, dwg, true, pw, 1.5e8, 64)) return 0;
  const unsigned na = px.grid.x * px.grid.y * px.grid.z, nb = pw.grid.x * pw.grid.y * pw.grid.z;
  const unsigned na8 = (na + 7u) & ~7u;
  const GldsDims dx{px.grid.x, px.grid.y, px.grid.z}, dw{pw.grid.x, pw.grid.y, pw.grid.z};
  const int lds = px.lds_bytes > pw.lds_bytes ? px.lds_bytes : pw.lds_bytes;
  hipEvent_t ev = (hipEvent_t)c->attach_event;   // ffh_event_record_with_next_linear_bwd: this launch is the call's last kernel on s
  const bool scatter = c->scatter_map && c->scatter_ncols == dxg.N && dxg.epi == EPI_STORE;
  if (scatter) {
    // ffh_linear_bwd_set_dx_scatter: the data gradient goes where a Concat backward would copy it; its event rides on this launch
    dxg.colmap = (const ffh_col_dest*)c->scatter_map;
    if (c->scatter_event && !ev) ev = (hipEvent_t)c->scatter_event;
  }
  // Inside a stream capture a launch's stop event is NOT a captured event-record node: the graph would carry no edge
  // from this kernel to whoever waits on the event.  There the event is recorded by an ordinary hipEventRecord behind
  // the launch (a captured node) instead.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (ev && hipStreamIsCapturing(as_stream(s), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  hipEvent_t ev_after = nullptr;
  if (ev && cap != hipStreamCaptureStatusNone) { ev_after = ev; ev = nullptr; }
  // more workgroups than CUs: two stages (65 KB) so that two workgroups share a CU
  static const int forced_stages = FFH_LAB_INT("FFH_GLDS_BWD_STAGES", 0);   // A/B switch (tools/ab.sh)
  const bool two = forced_stages ? forced_stages == 2 : (na8 + nb) > (unsigned)c->num_cus;
  const int lds2 = 2 * 128 * 256 + 1024;
#define FFH_DUAL(BMV, NS, LDSV)                                                                                                   \
  {                                                                                                                               \
    auto kern = gemm_glds_bwd_kernel<BMV, NS>;                                                                                    \
    static const bool ok = glds_set_lds(kern, NS * 128 * 256 + 1024);                                                             \
    if (!ok) return 0;                                                                                                            \
    if (ev) hipExtLaunchKernelGGL(kern, dim3(na8 + nb), dim3(kGldsWaves * 64), LDSV, as_stream(s), nullptr, ev, 0, dxg, dx, na8, dwg, dw); \
    else hipLaunchKernelGGL(kern, dim3(na8 + nb), dim3(kGldsWaves * 64), LDSV, as_stream(s), dxg, dx, na8, dwg, dw);              \
  }
  if (px.bm == 64) { if (two) FFH_DUAL(64, 2, lds2) else FFH_DUAL(64, 3, lds) }
  else { if (two) FFH_DUAL(32, 2, lds2) else FFH_DUAL(32, 3, lds) }
#undef FFH_DUAL
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, "linear_bwd dx+dw (lds-dma, one launch)");
  { char tok[96]; snprintf(tok, sizeof tok, "linear_bwd dx+dw|glds_dual_%dx64_s%d|splitk=%d", px.bm, two ? 2 : 3, dwg.splitk); ffh_route_add(c, tok); }
  if (scatter) c->scatter_used = 1;         // only once the launch that scatters is really in the stream
  if (ev_after) {
    e = hipEventRecord(ev_after, as_stream(s));
    if (e != hipSuccess) return ffh_fail_hip(c, e, "linear_bwd dx+dw: event record (capturing)");
    ev = ev_after;
  }
  if (ev && ev == (hipEvent_t)c->attach_event) c->attach_event = nullptr;   // signalled behind this kernel: no second record by the caller
  return 1;
}

// dy <- dy * act'(y) in place, and db[o] += sum_b dy[b][o]; one pass over dy.
// 256 threads = TX column-threads x TY row-threads; a thread owns VEC adjacent columns (16-B
// accesses when the layout allows), walks its rows with the partial column sums in registers,
// the TY partials of a column meet in LDS and one global atomic per column per workgroup is left.
template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ db, int out, int64_t batch, int rows_per_block, int act,
                                                           int tx_count) {
  ffh_kernel_prio();
  __shared__ float s_red[256 * VEC];
  const int TX = tx_count, TY = 256 / tx_count;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cols_v = (out + VEC - 1) / VEC;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < batch ? r0 + rows_per_block : batch;
  for (int c0 = 0; c0 < cols_v; c0 += TX) {      // uniform trip count: the LDS reduction below has barriers
    const int c = c0 + tx;
    const bool live = c < cols_v;
    float part[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) part[v] = 0.0f;
    // four rows per thread in flight (the loads of a row used to wait for the row before: 8 dependent round trips per thread
    // at the 128-wide layer, 60-100 us for a 33 MB pass); the column sums keep their row order
    constexpr int UR = 4;
    for (int64_t rb = r0 + ty; live && rb < r1; rb += (int64_t)TY * UR) {
      float d[UR][VEC], yo[UR][VEC];
#pragma unroll
      for (int u = 0; u < UR; u++) {
        const int64_t r = rb + (int64_t)u * TY;
        if (r >= r1) continue;
        const float* dp = dy + r * lddy + (int64_t)c * VEC;
        const float* yp = y + r * ldy + (int64_t)c * VEC;
        if (VEC == 4) {
          const float4 dv = *reinterpret_cast<const float4*>(dp);
          d[u][0] = dv.x; d[u][1] = dv.y; d[u][2] = dv.z; d[u][3] = dv.w;
          if (act != FFH_AC_MODE_NONE) { const float4 yv = *reinterpret_cast<const float4*>(yp); yo[u][0] = yv.x; yo[u][1] = yv.y; yo[u][2] = yv.z; yo[u][3] = yv.w; }
        } else {
          d[u][0] = dp[0];
          if (act != FFH_AC_MODE_NONE) yo[u][0] = yp[0];
        }
      }
#pragma unroll
      for (int u = 0; u < UR; u++) {
        const int64_t r = rb + (int64_t)u * TY;
        if (r >= r1) continue;
        float* dp = dy + r * lddy + (int64_t)c * VEC;
        if (act == FFH_AC_MODE_RELU) {
#pragma unroll
          for (int v = 0; v < VEC; v++) d[u][v] = (yo[u][v] > 0.0f) ? d[u][v] : 0.0f;
        } else if (act == FFH_AC_MODE_SIGMOID) {
#pragma unroll
          for (int v = 0; v < VEC; v++) d[u][v] = d[u][v] * yo[u][v] * (1 - yo[u][v]);
        }
        if (act != FFH_AC_MODE_NONE) {
          if (VEC == 4) *reinterpret_cast<float4*>(dp) = make_float4(d[u][0], d[u][1], d[u][2], d[u][3]);
          else dp[0] = d[u][0];
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) part[v] += d[u][v];
      }
    }
    if (db) {
      if (TY == 1) {
        if (live) {
#pragma unroll
          for (int v = 0; v < VEC; v++) atomicAdd(&db[c * VEC + v], part[v]);
        }
      } else {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; v++) s_red[(ty * TX + tx) * VEC + v] = part[v];
        __syncthreads();
        if (ty == 0 && live) {
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            float sum = 0.0f;
            for (int q = 0; q < TY; q++) sum += s_red[(q * TX + tx) * VEC + v];
            atomicAdd(&db[c * VEC + v], sum);
          }
        }
      }
    }
  }
}

// Batched GEMMs whose matrices are a fraction of a tile (the pairwise-dot interaction: 27 x 128 times 128 x 27 per
// sample, and its two backward products with k = 27): one WAVE per (batch item, 32x32 output tile), operands straight
// from global memory into the MFMA registers -- a tile is read once by one wave, so LDS staging would only add a
// round trip, and a 64x64 workgroup tile would be 80 % padding.  Lane (row r, half h) of v_mfma_f32_32x32x2_f32
// loads A(m0 + r, k + h) and B(n0 + r, k + h); eight k-steps of loads are issued before their MFMAs.
__global__ __launch_bounds__(256) void bmm_small_kernel(const GemmArgs g, const int tiles_m, const int tiles_n, const int64_t ntiles) {
  ffh_kernel_prio();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int per_batch = tiles_m * tiles_n;
  const int64_t bz = tile / per_batch;
  const int t2 = (int)(tile - bz * per_batch);
  const int m0 = (t2 / tiles_n) * 32, n0 = (t2 % tiles_n) * 32;
  const float* A = g.A + bz * g.bsA + (int64_t)(m0 + lr) * g.sAm;
  const float* B = g.B + bz * g.bsB + (int64_t)(n0 + lr) * g.sBn;
  const bool a_ok = m0 + lr < g.M, b_ok = n0 + lr < g.N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int k = k0 + 2 * u + lh;
      a[u] = (a_ok && k < g.K) ? A[(int64_t)k * g.sAk] : 0.0f;
      b[u] = (b_ok && k < g.K) ? B[(int64_t)k * g.sBk] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  const int n = n0 + lr;
  if (n < g.N) {
    float* C = g.C + bz * g.bsC;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= g.M) continue;
      float* cp = C + (int64_t)m * g.ldc + n;
      if (g.epi == EPI_STORE) *cp = acc[r];
      else *cp = *cp + acc[r];
    }
  }
}

template <bool AKC, bool BKC, bool FUSE_DY = false, bool CMAP = false>
int launch_gemm(ffh_ctx* c, GemmArgs& g, int64_t batch, ffh_stream s, const char* name) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || batch <= 0) return FFH_OK;
  if (batch > 1 && !FUSE_DY && g.epi != EPI_ATOMIC && !g.bias && !g.mask && g.act == FFH_AC_MODE_NONE &&
      (g.M <= 48 || g.N <= 48) && g.K <= 1024 && (int64_t)g.M * g.N <= 64 * 256) {
    // many matrices far smaller than a workgroup tile: one wave per 32x32 output tile, no LDS
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    const int64_t ntiles = (int64_t)tm * tn * batch;
    if ((ntiles + 3) / 4 <= 0x7fffffffLL) {
      hipLaunchKernelGGL(bmm_small_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, as_stream(s), g, tm, tn, ntiles);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return ffh_fail_hip(c, e, name);
      return FFH_OK;
    }
  }
  // tile choice: 128x128 when that alone fills the chip twice; 64x64 when it yields >= 2 workgroups
  // per CU; else one 32x32 tile per workgroup with the four waves splitting K
  const int64_t tiles128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * batch;
  const int64_t tiles64 = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64) * batch;
  int cfg;   // 0: 128x128, 1: 64x64, 2: 32x32 split-wave
  if (g.epi == EPI_ATOMIC) {
    // split-K supplies the parallelism here, so the tile follows the amount of work (measured on MI355X):
    // big tiles for deep/wide products (arithmetic intensity), the split-wave form for the skinny DLRM layers
    const double work = (double)g.M * g.N * g.K;
    cfg = work >= 3e9 && g.M >= 128 && g.N >= 128 ? 0 : (work >= 4e8 ? 1 : 2);
  } else if (tiles128 >= 2 * c->num_cus && g.M >= 128 && g.N >= 128) cfg = 0;
  else if (tiles64 >= 2 * c->num_cus || g.K < 64) cfg = 1;
  else cfg = 2;
  static const int forced = FFH_LAB_INT("FFH_GEMM_CFG", -1);   // A/B switch (tools/gemm_tune.py)
  if (forced >= 0 && forced <= 5 && (forced <= 2 || cfg == 0)) cfg = forced;
  if (CMAP) cfg = (cfg == 0 || cfg == 3 || cfg == 4 || cfg == 5) ? 0 : 1;     // the column-map epilogue exists for the two plain tile shapes
  const int BMv = cfg == 3 ? 256 : (cfg == 5 ? 128 : (cfg == 0 || cfg == 4 ? 128 : (cfg == 1 ? 64 : 32)));
  const int BNv = cfg == 3 ? 128 : (cfg == 5 ? 256 : BMv);
  const int gx = (g.N + BNv - 1) / BNv, gy = (g.M + BMv - 1) / BMv;
  int gz = (int)batch;
  g.splitk = 1;
  g.k_per_split = g.K;
  if (g.epi == EPI_ATOMIC && c->deterministic) g.epi = EPI_ADD;      // one workgroup per output tile adds its whole-K sum: no atomics
  if (g.epi == EPI_ATOMIC) {
    // split K over workgroups so that about two of them per CU are in flight; each split is a multiple of kSplitGran
    const int64_t tiles = (int64_t)gx * gy;
    int want = (int)((2LL * c->num_cus + tiles - 1) / tiles);
    const int max_split = (g.K + 4 * kSplitGran - 1) / (4 * kSplitGran);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int kps = (g.K + want - 1) / want;
    kps = (kps + 2 * kSplitGran - 1) / (2 * kSplitGran) * (2 * kSplitGran);
    g.k_per_split = kps;
    g.splitk = (g.K + kps - 1) / kps;
    if (g.splitk <= 1) { g.splitk = 2; }       // keeps blockIdx.z meaning "split" (second split is empty)
    gz = g.splitk;
    if (batch != 1) return ffh_fail(c, FFH_ERR_BAD_ARG, "gemm: split-K with a batch");
  }
  if (gy > 65535 || gz > 65535) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "gemm: grid too large");
  dim3 grid(gx, gy, gz);
  g.tnx = g.tny = g.tnz = 0;
  {
    // persistent form (see gemm_f32_kernel): launches of more tiles than the chip holds at once
    static const int persist = FFH_LAB_INT("FFH_GEMM_PERSIST", 0);   // A/B switch: resident workgroups per CU, 0 = off
    const int64_t total = (int64_t)gx * gy * gz;
    if (persist > 0 && total > (int64_t)persist * c->num_cus && total < (1LL << 31)) {
      g.tnx = (unsigned)gx; g.tny = (unsigned)gy; g.tnz = (unsigned)gz;
      grid = dim3((unsigned)(persist * c->num_cus), 1, 1);
    }
  }
  if (cfg == 3) hipLaunchKernelGGL((gemm_f32_kernel<256, 128, 16, AKC, BKC, false, FUSE_DY>), grid, dim3(256), 0, as_stream(s), g);
  else if (cfg == 4) hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 32, AKC, BKC, false, FUSE_DY>), grid, dim3(256), 0, as_stream(s), g);
  else if (cfg == 5) hipLaunchKernelGGL((gemm_f32_kernel<128, 256, 16, AKC, BKC, false, FUSE_DY>), grid, dim3(256), 0, as_stream(s), g);
  else if (cfg == 0) hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 16, AKC, BKC, false, FUSE_DY, CMAP>), grid, dim3(256), 0, as_stream(s), g);
  else if (cfg == 1) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 32, AKC, BKC, false, FUSE_DY, CMAP>), grid, dim3(256), 0, as_stream(s), g);
  else hipLaunchKernelGGL((gemm_f32_kernel<32, 32, 64, AKC, BKC, true, FUSE_DY>), grid, dim3(256), 0, as_stream(s), g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  { char tok[96]; snprintf(tok, sizeof tok, "%s|f32_%dx%d_cfg%d|splitk=%d", name, BMv, BNv, cfg, g.splitk); ffh_route_add(c, tok); }
  return FFH_OK;
}


// =============================================================================================
// Linear with a handful of outputs (the click-probability layer of the DLRM top MLP: out = 1).  GEMV-shaped and
// HBM-bound: a 32x32 MFMA tile would be 97 % padding and the layer would cost three or four launches; here the
// forward is one wave per sample row and the whole backward (activation gradient in place, db, dW, dX) ONE launch.
// =============================================================================================
constexpr int kSkinnyMaxOut = 4;
constexpr int kSkinnyMaxIn = 1024;      // backward: a thread keeps <= 4 columns x kSkinnyMaxOut partial dW sums

__global__ __launch_bounds__(256) void linear_skinny_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ w, const float* __restrict__ bias, int in, int out,
                                                                int64_t batch, int act) {
  ffh_kernel_prio();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < batch; b += (int64_t)gridDim.x * 4) {
    float acc[kSkinnyMaxOut] = {0.f, 0.f, 0.f, 0.f};
    const float* xr = x + b * ldx;
    for (int i = lane; i < in; i += 64) {
      const float xv = xr[i];
#pragma unroll
      for (int o = 0; o < kSkinnyMaxOut; o++)
        if (o < out) acc[o] = __fmaf_rn(xv, w[(int64_t)o * in + i], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < kSkinnyMaxOut; o++)
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) acc[o] += __shfl_down(acc[o], d);
    if (lane == 0)
      for (int o = 0; o < out; o++) y[b * ldy + o] = act_apply(acc[o] + (bias ? bias[o] : 0.0f), act);
  }
}

// Linear with a handful of INPUTS (the 13 dense features in front of DLRM's bottom MLP): k is shorter than one LDS k-tile, so
// the tiled kernels spend their time on padding and staging.  Here a wave owns a 32 x 32 output tile and takes both
// operands straight from global memory (in <= 16 values per row: at most eight MFMAs), bias + activation, one store pass.
__global__ __launch_bounds__(512) void linear_thin_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ w, const float* __restrict__ bias, int in, int out,
                                                              int64_t batch, int act) {
  ffh_kernel_prio();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int n0 = (blockIdx.y * 8 + wave) * 32;
  if (n0 >= out) return;
  const bool xok = b0 + r < batch, wok = n0 + r < out;
  float av[8], bv[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int k = 2 * q + h;
    av[q] = (xok && k < in) ? x[(b0 + r) * ldx + k] : 0.0f;
    bv[q] = (wok && k < in) ? w[(int64_t)(n0 + r) * in + k] : 0.0f;
  }
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; v++) acc[v] = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; q++)
    if (2 * q < in) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
  if (!wok) return;
  const float bs = bias ? bias[n0 + r] : 0.0f;
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int64_t i = b0 + 8 * (v >> 2) + 4 * h + (v & 3);
    if (i < batch) y[i * ldy + n0 + r] = act_apply(acc[v] + bs, act);
  }
}

// The same layer as an HBM-write-bound stream (the 13 -> 512 layer at 32768 samples writes 67 MB and reads 1.7 MB): a wave owns
// a ROW and 256 consecutive columns -- lane l the four columns 4l .. 4l+3, so a row segment leaves as ONE 1-KiB store
// instruction; the <= 16 inputs of the row are wave-uniform (staged through LDS, below), the lane's 4 x in weights stay in registers for
// all of the workgroup's rows.  52 FMAs per 16 bytes written: the VALU work is a tenth of the store time.  Each output is
// the ascending-k fmaf chain from 0, then + bias, then the activation -- the oracle's order, bit for bit.
constexpr int kThinRowsPerWg = 128;
__global__ __launch_bounds__(256) void linear_thin_fwd_rows_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                                   const float* __restrict__ w, const float* __restrict__ bias, int in, int out,
                                                                   int64_t batch, int act) {
  ffh_kernel_prio();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = (blockIdx.y * 64 + lane) * 4;
  const bool live = n0 < out;                    // out % 4 == 0 (host check)
  // the workgroup's 256 x in slice of w is contiguous: one coalesced sweep into LDS, transposed ([k][column], rows of 260 floats),
  // from where a lane takes its 4 columns of every k with one ds_read_b128 -- read straight from memory this was 52 loads per
  // lane with 64 lanes 208 bytes apart, 64 cache lines per instruction (32.5 -> 30 us; 67 MB written: the stream alone would be ~13)
  __shared__ __attribute__((aligned(16))) float wT[16][260];
  {
    const int c0 = blockIdx.y * 256;
    const int ncol = out - c0 < 256 ? out - c0 : 256;
    for (int e = threadIdx.x; e < 16 * 256; e += 256) wT[e >> 8][e & 255] = 0.0f;
    __syncthreads();
    const float* wb = w + (int64_t)c0 * in;
    for (int e = threadIdx.x; e < ncol * in; e += 256) wT[e % in][e / in] = wb[e];
    __syncthreads();
  }
  float wr[16][4];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const float4 t = *reinterpret_cast<const float4*>(&wT[k][4 * lane]);
    wr[k][0] = t.x; wr[k][1] = t.y; wr[k][2] = t.z; wr[k][3] = t.w;
  }
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && bias) bs = *reinterpret_cast<const float4*>(bias + n0);
  // the workgroup's 128 input rows go through LDS in one coalesced sweep (rows padded to 16 floats): a wave then takes a row's
  // inputs with four broadcast ds_read_b128 instead of a chain of dependent scalar loads per row (39 -> 32.5 us at 32768 samples)
  __shared__ __attribute__((aligned(16))) float xs_l[kThinRowsPerWg][16];
  const int64_t r0 = (int64_t)blockIdx.x * kThinRowsPerWg;
  const int nrows = (int)(batch - r0 < kThinRowsPerWg ? batch - r0 : kThinRowsPerWg);
  for (int i = threadIdx.x; i < kThinRowsPerWg * 16; i += 256) {
    const int rr = i >> 4, k = i & 15;
    xs_l[rr][k] = (rr < nrows && k < in) ? x[(r0 + rr) * ldx + k] : 0.0f;
  }
  __syncthreads();
  for (int rr = wave; rr < nrows; rr += 4) {
    const float4* xr = reinterpret_cast<const float4*>(xs_l[rr]);
    const float4 q0 = xr[0], q1 = xr[1], q2 = xr[2], q3 = xr[3];
    const float xs[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < in) {
        a0 = __fmaf_rn(xs[k], wr[k][0], a0); a1 = __fmaf_rn(xs[k], wr[k][1], a1);
        a2 = __fmaf_rn(xs[k], wr[k][2], a2); a3 = __fmaf_rn(xs[k], wr[k][3], a3);
      }
    }
    if (bias) { a0 = a0 + bs.x; a1 = a1 + bs.y; a2 = a2 + bs.z; a3 = a3 + bs.w; }
    if (live) *reinterpret_cast<float4*>(y + (r0 + rr) * ldy + n0) = make_float4(act_apply(a0, act), act_apply(a1, act), act_apply(a2, act), act_apply(a3, act));
  }
}

struct SkinnyBwdArgs {
  const float* x;  float* dx;  const float* y;  float* dy;  const float* w;  float* dw;  float* db;
  int64_t ldx, lddx, ldy, lddy, batch;
  int in, out, rows_per_block;
  int act;          // activation whose derivative is applied to dy here (NONE: dy is taken as it is)
  int write_back;   // store the activation gradient into dy (the reference mutates dy in place)
  int do_db, do_dw, do_dx, dx_overwrite, mask_by_x;
  // MSE loss step folded in (ffh_linear_bwd_mse): dy is not read but made here from y and label; metrics as metrics_kernel
  const float* label;  float loss_scale;  ffh_perf_metrics* perf;  int metrics_flags;
  unsigned short* dx16;       // tensor-op mode: the bf16 twin of dx (ffh_ctx_bf16_mirror_set), written beside it, or null
  // dW / db without atomic chains: workgroup b leaves its partial row [out * in + out] at ws + b * ws_stride, the last one to arrive
  // (ws_cnt) adds the rows up in block order and adds the total to dw / db.  Null: one atomic per weight per workgroup.
  float* ws;  unsigned* ws_cnt;  int ws_stride;
};

// NC: 16-byte column chunks per lane (in <= 256 * NC), NO: output slots kept in registers (out <= NO),
// RPW: rows per wave-instruction -- a row of in = 256 / RPW floats fills 64 / RPW lanes, so RPW rows go side by side
// (lane = rsub * (64 / RPW) + chunk) instead of leaving three quarters of the wave idle on the 64-wide layer
template <int NC, int NO, int RPW = 1>
__global__ __launch_bounds__(256) void linear_skinny_bwd_kernel(const SkinnyBwdArgs a) {
  ffh_kernel_prio();
  static_assert(RPW == 1 || NC == 1, "row packing is for rows narrower than a wave");
  constexpr int LPR = 64 / RPW;
  extern __shared__ float s_dz[];                       // [rows_per_block][out], then the cross-wave dW reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * a.rows_per_block;
  const int rows = (int)((a.batch - b0) < a.rows_per_block ? (a.batch - b0) : a.rows_per_block);
  if (rows <= 0) return;
  // 1. activation gradient of this block's rows: reluBackward [ref: src/runtime/cuda_helper.cu:71-78] /
  //    sigmoid_backward [ref: src/ops/linear.cu:600-607], once per element, then shared through LDS
  float mse_s = 0.f, rmse_s = 0.f, mae_s = 0.f;
  int m_all = 0, m_correct = 0;
  for (int e = tid; e < rows * a.out; e += 256) {
    const int r = e / a.out, o = e - r * a.out;
    float d;
    if (a.label) {
      // mean_squared_error_avg_loss_backward + scale [ref: src/loss_functions/loss_functions.cu:65-76,160-166], same rounding as metrics_kernel
      d = __fmaf_rn(a.loss_scale - 0.0f, a.y[(b0 + r) * a.ldy + o] - a.label[(b0 + r) * a.out + o], 0.0f);
      if (o == 0) {   // per-sample metrics [ref: src/metrics_functions/metrics_functions.cu:108-173], as metrics_kernel computes them
        const float* lg = a.y + (b0 + r) * a.ldy;
        const float* lb = a.label + (b0 + r) * a.out;
        m_all += 1;
        if (a.metrics_flags & 1) {
          if (a.out == 1) { m_all += 1; m_correct += 1; }
          else {
            float max_val = 0.0f; int my = -1, tr = -1;
            for (int i = 0; i < a.out; i++) {
              const float lv = lg[i];
              if (my == -1 || lv > max_val) { max_val = lv; my = i; }
              if (lb[i] > 0.9f) tr = i;
            }
            if (tr == my) m_correct += 1;
          }
        }
        if (a.metrics_flags & (2 | 4 | 8)) {
          float mse = 0.f, mae = 0.f;
          for (int i = 0; i < a.out; i++) {
            const float diff = lg[i] - lb[i];
            mse = __fmaf_rn(diff, diff, mse);
            mae += fabsf(diff);
          }
          mse_s += mse; rmse_s += sqrtf(mse); mae_s += mae;
        }
      }
    } else {
      d = a.dy[(b0 + r) * a.lddy + o];
    }
    if (a.act == FFH_AC_MODE_RELU) d = a.y[(b0 + r) * a.ldy + o] > 0.0f ? d : 0.0f;
    else if (a.act == FFH_AC_MODE_SIGMOID) { const float yo = a.y[(b0 + r) * a.ldy + o]; d = d * yo * (1 - yo); }
    if (a.write_back) a.dy[(b0 + r) * a.lddy + o] = d;
    s_dz[e] = d;
  }
  if (a.label) {   // one atomic per counter per workgroup (uniform branch: label is a kernel argument)
    __shared__ float s_mf[3][4];
    __shared__ int s_mi[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mse_s += __shfl_down(mse_s, o); rmse_s += __shfl_down(rmse_s, o); mae_s += __shfl_down(mae_s, o);
      m_all += __shfl_down(m_all, o); m_correct += __shfl_down(m_correct, o);
    }
    if (lane == 0) { s_mf[0][wave] = mse_s; s_mf[1][wave] = rmse_s; s_mf[2][wave] = mae_s; s_mi[0][wave] = m_all; s_mi[1][wave] = m_correct; }
    __syncthreads();
    if (tid == 0) {
      const int al = s_mi[0][0] + s_mi[0][1] + s_mi[0][2] + s_mi[0][3];
      const int co = s_mi[1][0] + s_mi[1][1] + s_mi[1][2] + s_mi[1][3];
      if (al) atomicAdd(&a.perf->train_all, al);
      if (co) atomicAdd(&a.perf->train_correct, co);
      if (a.metrics_flags & 2) atomicAdd(&a.perf->mse_loss, (s_mf[0][0] + s_mf[0][1]) + (s_mf[0][2] + s_mf[0][3]));
      if (a.metrics_flags & 4) atomicAdd(&a.perf->rmse_loss, (s_mf[1][0] + s_mf[1][1]) + (s_mf[1][2] + s_mf[1][3]));
      if (a.metrics_flags & 8) atomicAdd(&a.perf->mae_loss, (s_mf[2][0] + s_mf[2][1]) + (s_mf[2][2] + s_mf[2][3]));
    }
  }
  __syncthreads();
  if (a.do_db && a.db && tid < a.out) {
    float sum = 0.f;
    for (int r = 0; r < rows; r++) sum += s_dz[r * a.out + tid];
    if (a.ws) __hip_atomic_store(a.ws + (int64_t)blockIdx.x * a.ws_stride + a.out * a.in + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicAdd(&a.db[tid], sum);
  }
  // 2. a wave takes rows wave, wave + 4, ...; a lane owns the 16-byte column chunks lane, lane + 64, ... of every row
  //    (in <= 1024: at most 4 chunks), so a row is one coalesced pass; four rows are in flight per wave
  const int nch = a.in / 4;
  const int rsub = lane / LPR, lch = lane - rsub * LPR;          // RPW == 1: rsub = 0, lch = lane
  float4 wv[NC][NO];
  float4 dwacc[NC][NO];
#pragma unroll
  for (int c = 0; c < NC; c++)
#pragma unroll
    for (int o = 0; o < NO; o++) {
      const int ch = lch + 64 * c;
      wv[c][o] = (ch < nch && o < a.out && a.do_dx) ? reinterpret_cast<const float4*>(a.w + (int64_t)o * a.in)[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
      dwacc[c][o] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  constexpr int U = NO > 4 ? 4 : 16 / NC;   // rows in flight per lane: U x NC 16-byte loads (fewer when NO slots fill the registers)
  for (int r0 = wave * RPW + rsub; r0 < rows + rsub; r0 += 4 * RPW * U) {
    float4 xv[U][NC];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int r = r0 + 4 * RPW * u, ch = lch + 64 * c;
        xv[u][c] = (r < rows && ch < nch) ? reinterpret_cast<const float4*>(a.x + (b0 + r) * a.ldx)[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = r0 + 4 * RPW * u;
      if (r >= rows) continue;
      float dz[NO];
#pragma unroll
      for (int o = 0; o < NO; o++) dz[o] = o < a.out ? s_dz[r * a.out + o] : 0.0f;
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int ch = lch + 64 * c;
        if (ch >= nch) continue;
        const float4 x4 = xv[u][c];
        if (a.do_dw) {
#pragma unroll
          for (int o = 0; o < NO; o++) {
            dwacc[c][o].x = __fmaf_rn(dz[o], x4.x, dwacc[c][o].x); dwacc[c][o].y = __fmaf_rn(dz[o], x4.y, dwacc[c][o].y);
            dwacc[c][o].z = __fmaf_rn(dz[o], x4.z, dwacc[c][o].z); dwacc[c][o].w = __fmaf_rn(dz[o], x4.w, dwacc[c][o].w);
          }
        }
        if (a.do_dx) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int o = 0; o < NO; o++) {
            v.x = __fmaf_rn(dz[o], wv[c][o].x, v.x); v.y = __fmaf_rn(dz[o], wv[c][o].y, v.y);
            v.z = __fmaf_rn(dz[o], wv[c][o].z, v.z); v.w = __fmaf_rn(dz[o], wv[c][o].w, v.w);
          }
          if (a.mask_by_x) {
            v.x = x4.x > 0.0f ? v.x : 0.0f; v.y = x4.y > 0.0f ? v.y : 0.0f; v.z = x4.z > 0.0f ? v.z : 0.0f; v.w = x4.w > 0.0f ? v.w : 0.0f;
          }
          float4* p = reinterpret_cast<float4*>(a.dx + (b0 + r) * a.lddx) + ch;
          if (!a.dx_overwrite) { const float4 q = *p; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          *p = v;
          if (a.dx16) {
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            const bf16x4_t t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *reinterpret_cast<bf16x4_t*>(a.dx16 + (b0 + r) * a.lddx + 4 * ch) = t;
          }
        }
      }
    }
  }
  if (RPW > 1 && a.do_dw) {
    // the RPW row slots of a lane's chunk meet in the slot-0 lanes
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) {
        dwacc[0][o].x += __shfl_xor(dwacc[0][o].x, m); dwacc[0][o].y += __shfl_xor(dwacc[0][o].y, m);
        dwacc[0][o].z += __shfl_xor(dwacc[0][o].z, m); dwacc[0][o].w += __shfl_xor(dwacc[0][o].w, m);
      }
  }
  if (a.do_dw) {
    // the four waves' partial sums meet in LDS, one atomic per weight per workgroup
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(s_dz);       // [wave][out][nch]
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int ch = lch + 64 * c;
        if (o < a.out && ch < nch && rsub == 0) red[((int64_t)wave * a.out + o) * nch + ch] = dwacc[c][o];
      }
    __syncthreads();
    typedef unsigned long long u64;
    for (int e = tid; e < a.out * nch; e += 256) {
      const float4 p0 = red[e], p1 = red[a.out * nch + e], p2 = red[2 * a.out * nch + e], p3 = red[3 * a.out * nch + e];
      const float4 s4 = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
      if (a.ws) {
        u64* q = reinterpret_cast<u64*>(a.ws + (int64_t)blockIdx.x * a.ws_stride + (int64_t)e * 4);
        __hip_atomic_store(q, ((u64)__float_as_uint(s4.y) << 32) | __float_as_uint(s4.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, ((u64)__float_as_uint(s4.w) << 32) | __float_as_uint(s4.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        float* d = a.dw + (int64_t)e * 4;                 // e = o * nch + ch  ->  dw[o][4 ch]
        atomicAdd(d + 0, s4.x); atomicAdd(d + 1, s4.y); atomicAdd(d + 2, s4.z); atomicAdd(d + 3, s4.w);
      }
    }
  }
  if (a.ws && (a.do_dw || (a.do_db && a.db))) {
    // The last workgroup to arrive adds the partial rows up -- in block order: the same bits run to run -- and adds the totals to dw /
    // db.  With one atomic per weight per workgroup every weight was the end of a chain of gridDim.x serialised adds (~0.1 us each:
    // 256 -> 1 at 32768 samples 33 us for dW + db in 256 workgroups, 58 in 512, while the dX stream alone wants 512: 13.9 us).
    // Cross-workgroup values travel as agent-scope relaxed atomics ordered by completion (vmcnt + barrier before the counter), as in
    // embedding.hip's folds; the counter is left at 0 for the next launch on this stream (also a replayed hipGraph node).
    __shared__ unsigned s_prev;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_prev = __hip_atomic_fetch_add(a.ws_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_prev != gridDim.x - 1) return;
    if (tid == 0) __hip_atomic_store(a.ws_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    typedef unsigned long long u64;
    const int ng = a.do_dw ? a.out * nch : 0;            // float4 groups of dW
    const int nb = (int)gridDim.x;
    float4* comb = reinterpret_cast<float4*>(s_dz);      // [part][group] when several threads share a group (ng <= 128)
    const int parts = ng > 0 && ng <= 128 ? 256 / ng : 1;      // threads per group (ng = 64 for 256 -> 1: 4)
    const int gpi = parts > 1 ? ng : 256;                       // groups per sweep of the workgroup
    __syncthreads();                                     // red[] is dead: s_dz is reused
    for (int g0 = 0; g0 < ng; g0 += gpi) {
      const int g = g0 + tid % gpi, part = tid / gpi;
      const bool live = g < ng && part < parts;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        const u64* q = reinterpret_cast<const u64*>(a.ws + (int64_t)g * 4);
        constexpr int UN = 8;
        for (int b = part; b < nb; b += parts * UN) {
          u64 lo[UN], hi[UN];
#pragma unroll
          for (int u = 0; u < UN; u++) {
            const int bb = b + u * parts;
            const u64* qq = q + (int64_t)(bb < nb ? bb : b) * (a.ws_stride / 2);
            lo[u] = __hip_atomic_load(qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hi[u] = __hip_atomic_load(qq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < UN; u++)
            if (b + u * parts < nb) {
              acc.x += __uint_as_float((unsigned)lo[u]); acc.y += __uint_as_float((unsigned)(lo[u] >> 32));
              acc.z += __uint_as_float((unsigned)hi[u]); acc.w += __uint_as_float((unsigned)(hi[u] >> 32));
            }
        }
      }
      if (parts > 1) {
        if (live) comb[part * gpi + (g - g0)] = acc;
        __syncthreads();
        if (live && part == 0)
          for (int pp = 1; pp < parts; pp++) { const float4 o4 = comb[pp * gpi + (g - g0)]; acc.x += o4.x; acc.y += o4.y; acc.z += o4.z; acc.w += o4.w; }
        __syncthreads();
      }
      if (live && part == 0) {
        float4* d = reinterpret_cast<float4*>(a.dw + (int64_t)g * 4);
        float4 o4 = *d; o4.x += acc.x; o4.y += acc.y; o4.z += acc.z; o4.w += acc.w; *d = o4;
      }
    }
    if (a.do_db && a.db) {
      __shared__ float s_dbw[4];
      for (int o = 0; o < a.out; o++) {
        float sum = 0.f;
        for (int b = tid; b < nb; b += 256) sum += __hip_atomic_load(a.ws + (int64_t)b * a.ws_stride + a.out * a.in + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) sum += __shfl_down(sum, dd);
        if (lane == 0) s_dbw[wave] = sum;
        __syncthreads();
        if (tid == 0) a.db[o] += (s_dbw[0] + s_dbw[1]) + (s_dbw[2] + s_dbw[3]);
        __syncthreads();
      }
    }
  }
}

// =============================================================================================
// Two narrow layers at the end of a chain (DLRM's bottom MLP ends 256 -> 64 -> 16): the UPPER layer's whole backward and the
// LOWER layer's data gradient in ONE launch.  As separate launches these are three latency-bound kernels (out <= 16
// one-launch backward, then the lower layer's dX GEMM) whose operands are a few KB per 32 samples; here a workgroup of eight
// waves owns 32 samples and keeps the 32 x in_u activation gradient between the layers in LDS:
//   g_l = (dy_u W_u) . act_l'(x_u)          MFMA, K = out_u <= 16      -> LDS and dy_l (the lower layer's premasked dy)
//   dW_u += dy_u^T x_u, db_u += sum dy_u    MFMA, K = the 32 samples   -> atomics (out_u x in_u values per workgroup)
//   dX_l = g_l W_l (. relu'(x_l))           MFMA, K = in_u <= 64, one 32-column tile per wave
// The lower layer's dW / db stay a GEMM over the whole batch (a per-workgroup partial of in_u x in_l values would cost
// more in atomics than the launch saves).
// =============================================================================================
struct PairBwdArgs {
  const float* xu; int64_t ldxu;                         // upper layer's input = lower layer's output  [B][in_u]
  float* dyu; int64_t lddyu; const float* yu; int64_t ldyu;   // upper layer's output gradient (activation gradient written back) and output
  const float* wu; float* dwu; float* dbu;               // W_u [out_u][in_u]
  int in_u, out_u, act_u;                                // act_u: NONE when dy_u is premasked
  const float* xl; int64_t ldxl;                         // lower layer's input  [B][in_l] (read for the mask only)
  float* dxl; int64_t lddxl;                             // lower layer's data gradient
  float* dyl; int64_t lddyl;                             // lower layer's output gradient, written premasked
  const float* wl;                                       // W_l [in_u][in_l]
  int in_l, act_l, dxl_overwrite, dxl_mask_by_x;
  int64_t batch;
};

__global__ __launch_bounds__(512) void linear_pair_bwd_kernel(const PairBwdArgs a) {
  ffh_kernel_prio();
  __shared__ float s_gu[32][17];                         // dy_u tile after its activation gradient
  __shared__ float s_gl[32][65];                         // g_l tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int rows = (int)((a.batch - b0) < 32 ? (a.batch - b0) : 32);
  {   // 1. the upper layer's activation gradient, once per element (in place, as the reference)
    const int i = tid >> 4, o = tid & 15;
    float d = 0.0f;
    if (i < rows && o < a.out_u) {
      d = a.dyu[(b0 + i) * a.lddyu + o];
      if (a.act_u == FFH_AC_MODE_RELU) d = a.yu[(b0 + i) * a.ldyu + o] > 0.0f ? d : 0.0f;
      else if (a.act_u == FFH_AC_MODE_SIGMOID) { const float yo = a.yu[(b0 + i) * a.ldyu + o]; d = d * yo * (1 - yo); }
      if (a.act_u != FFH_AC_MODE_NONE) a.dyu[(b0 + i) * a.lddyu + o] = d;
    }
    s_gu[i][o] = d;
  }
  __syncthreads();
  const int ntu = a.in_u / 32;                           // 32-column tiles of the upper layer's input (1 or 2)
  if (wave < ntu) {
    // 2a. g_l tile: A = dy_u (rows = samples, k = out_u padded to 16), B = W_u[k][n]
    const int n0 = wave * 32;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; v++) acc[v] = 0.0f;
    float bv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int k = 8 * h + q; bv[q] = k < a.out_u ? a.wu[(int64_t)k * a.in_u + n0 + r] : 0.0f; }
#pragma unroll
    for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_gu[r][8 * h + q], bv[q], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int i = 8 * (v >> 2) + 4 * h + (v & 3);
      float g = acc[v];
      if (i < rows) {
        if (a.act_l == FFH_AC_MODE_RELU) g = a.xu[(b0 + i) * a.ldxu + n0 + r] > 0.0f ? g : 0.0f;
        a.dyl[(b0 + i) * a.lddyl + n0 + r] = g;
      } else {
        g = 0.0f;
      }
      s_gl[i][n0 + r] = g;
    }
  } else if (wave < 2 * ntu) {
    // 2b. dW_u tile: A = dy_u^T (rows = out_u padded to 32, k = the 32 samples), B = x_u[k][n]
    const int n0 = (wave - ntu) * 32;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; v++) acc[v] = 0.0f;
    float bv[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { const int i = 16 * h + q; bv[q] = i < rows ? a.xu[(b0 + i) * a.ldxu + n0 + r] : 0.0f; }
#pragma unroll
    for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r < 16 ? s_gu[16 * h + q][r] : 0.0f, bv[q], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int m = 8 * (v >> 2) + 4 * h + (v & 3);
      if (m < a.out_u) atomicAdd(&a.dwu[(int64_t)m * a.in_u + n0 + r], acc[v]);
    }
  } else if (wave == 2 * ntu && a.dbu) {
    if (lane < a.out_u) {
      float sum = 0.0f;
      for (int i = 0; i < 32; i++) sum += s_gu[i][lane];
      atomicAdd(&a.dbu[lane], sum);
    }
  }
  __syncthreads();
  // 3. dX_l: one 32-column tile per wave; A = g_l (k = in_u), B = W_l[k][n]
  const int half = a.in_u / 2;                           // lane half h supplies k = half * h + q
  for (int n0 = wave * 32; n0 < a.in_l; n0 += 8 * 32) {
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; v++) acc[v] = 0.0f;
    for (int q0 = 0; q0 < half; q0 += 16) {
      float bv[16];
#pragma unroll
      for (int q = 0; q < 16; q++) bv[q] = a.wl[(int64_t)(half * h + q0 + q) * a.in_l + n0 + r];
#pragma unroll
      for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_gl[r][half * h + q0 + q], bv[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int i = 8 * (v >> 2) + 4 * h + (v & 3);
      if (i >= rows) continue;
      float g = acc[v];
      if (a.dxl_mask_by_x) g = a.xl[(b0 + i) * a.ldxl + n0 + r] > 0.0f ? g : 0.0f;
      float* p = a.dxl + (b0 + i) * a.lddxl + n0 + r;
      *p = a.dxl_overwrite ? g : *p + g;
    }
  }
}


// The same two narrow layers forward: y_l = act_l(x_l W_l^T + b_l) and y_u = act_u(y_l W_u^T + b_u) in ONE launch.  Eight
// waves own 32 samples: (mid / 32) output tiles x (8 / tiles) k-slices of the first product, operands straight from global
// memory as 16-byte pieces (both are k-contiguous; which k a lane half holds is free as long as A and B agree); the
// k-slices meet in LDS, bias + activation, y_l goes to memory and stays in LDS as the A operand of the second product.
struct PairFwdArgs {
  const float* xl; int64_t ldxl; const float* wl; const float* bl; int in_l, act_l;
  float* yl; int64_t ldyl; int mid;
  const float* wu; const float* bu; int out_u, act_u;
  float* yu; int64_t ldyu;
  int64_t batch;
};

__global__ __launch_bounds__(512) void linear_pair_fwd_kernel(const PairFwdArgs a) {
  ffh_kernel_prio();
  __shared__ float s_red[8 * 16 * 64];                   // [wave][reg][lane]: the k-slices of the first product
  __shared__ float s_yl[32][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int rows = (int)((a.batch - b0) < 32 ? (a.batch - b0) : 32);
  const int nt = a.mid / 32, ks = 8 / nt;                // output tiles, k-slices per tile
  const int tile = wave % nt, slice = wave / nt;
  const int kper = a.in_l / ks;                          // a multiple of 32 (checked by the caller)
  const int n0 = tile * 32;
  // second product's B operand (W_u[n][k], n < out_u): requested now, used last (wave 0)
  float4 wu4[8];
  const int half2 = a.mid / 2;                           // lane half h supplies k = half2 * h + q, q < half2 <= 32
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 8; q++)
      wu4[q] = (r < a.out_u && 4 * q < half2) ? *reinterpret_cast<const float4*>(a.wu + (int64_t)r * a.mid + half2 * h + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; v++) acc[v] = 0.0f;
  const float* xrow = a.xl + (b0 + r) * a.ldxl;
  const float* wrow = a.wl + (int64_t)(n0 + r) * a.in_l;
  const bool xok = r < rows;
  for (int k0 = slice * kper; k0 < (slice + 1) * kper; k0 += 32) {
    float4 av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      av[q] = xok ? *reinterpret_cast<const float4*>(xrow + k0 + 16 * h + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[q] = *reinterpret_cast<const float4*>(wrow + k0 + 16 * h + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int v = 0; v < 16; v++) s_red[(wave * 16 + v) * 64 + lane] = acc[v];
  __syncthreads();
  {
    // wave (tile, slice) finishes registers slice * (16 / ks) ... of its tile: ((s0 + s1) + s2) + ..., bias, activation
    const int rpw = 16 / ks;
    const float bias = a.bl ? a.bl[n0 + r] : 0.0f;
    for (int q = 0; q < rpw; q++) {
      const int v = slice * rpw + q;
      float sum = s_red[((0 * nt + tile) * 16 + v) * 64 + lane];
      for (int s2 = 1; s2 < ks; s2++) sum += s_red[((s2 * nt + tile) * 16 + v) * 64 + lane];
      const int i = 8 * (v >> 2) + 4 * h + (v & 3);
      const float y = act_apply(sum + bias, a.act_l);
      s_yl[i][n0 + r] = i < rows ? y : 0.0f;
      if (i < rows) a.yl[(b0 + i) * a.ldyl + n0 + r] = y;
    }
  }
  __syncthreads();
  if (wave == 0) {
    f32x16 acc2;
#pragma unroll
    for (int v = 0; v < 16; v++) acc2[v] = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (4 * q < half2) {
        const float* ap = &s_yl[r][half2 * h + 4 * q];
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], wu4[q].x, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[1], wu4[q].y, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2], wu4[q].z, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[3], wu4[q].w, acc2, 0, 0, 0);
      }
    }
    if (r < a.out_u) {
      const float bias = a.bu ? a.bu[r] : 0.0f;
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int i = 8 * (v >> 2) + 4 * h + (v & 3);
        if (i < rows) a.yu[(b0 + i) * a.ldyu + r] = act_apply(acc2[v] + bias, a.act_u);
      }
    }
  }
}

bool act_ok(int act) { return act == FFH_AC_MODE_NONE || act == FFH_AC_MODE_RELU || act == FFH_AC_MODE_SIGMOID; }
// forward also serves GELU; the reference's backward does not [ref: src/ops/linear.cu:632-635 asserts NONE / RELU / SIGMOID]
bool act_ok_fwd(int act) { return act_ok(act) || act == FFH_AC_MODE_GELU; }

}  // namespace

extern "C" {

// the reduction depth a caller pads x / w to so that the layer meets the persistent kernels' contract (include/ff_hip.h)
int ffh_linear_fast_in_dim(int in, int out) {
  if (in <= 0 || out <= 0) return in;
  if (in % 64 == 0 || in < 256 || out % 128 != 0) return in;
  return (in + 63) / 64 * 64;
}

int ffh_linear_fwd(ffh_ctx* c, const float* x, int64_t ldx, float* y, int64_t ldy, const float* w, const float* bias,
                   int in, int out, int64_t batch, int act, ffh_stream s) {
  FFH_REQUIRE(c, in > 0 && out > 0 && batch >= 0 && ldx >= in && ldy >= out, "linear_fwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x && y && w), "linear_fwd: null pointer");
  FFH_REQUIRE(c, batch < (1LL << 31), "linear_fwd: batch too large");
  if (!act_ok_fwd(act)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_fwd: activation not supported (NONE, RELU, SIGMOID, GELU)");
  ffh_route_clear(c);
  if (batch == 0) return FFH_OK;
  if (out <= kSkinnyMaxOut) {
    hipLaunchKernelGGL(linear_skinny_fwd_kernel, dim3(ffh_grid(batch, 4, 2048)), dim3(256), 0, as_stream(s), x, ldx, y, ldy, w, bias, in, out,
                       batch, act);
    FFH_LAUNCH_CHECK(c, "linear_skinny_fwd_kernel");
    ffh_route_add(c, "linear_fwd|skinny");
    return FFH_OK;
  }
  static const int no_thin = FFH_LAB_INT("FFH_NO_THIN", 0);   // A/B switch (tools/ab.sh)
  // (rows kernel from 8192 samples up: it keeps 128 rows per workgroup to amortise its weight registers, so a 2048-sample launch
  //  would be 32 workgroups -- the MFMA form below is the faster one there: Kaggle step 208 vs 185 us)
  // Round 4 re-measured the three forms of the 13 -> 512 layer alone (profiles/r04_ab_schedule.txt): at 32768 samples the rows kernel
  // 31.8 us, the MFMA thin kernel 25.4, the ordinary GEMM path 23.4; at 8192: 23.9 / 9.5 / 14.0; at 4096: 7.7 / 7.5 / 8.2 -- the rows
  // kernel is off (A/B: FFH_THIN_ROWS_MIN_BATCH), the thin kernel serves below 16384 samples (FFH_THIN_MAX_BATCH)
  static const int64_t thin_rows_min = FFH_LAB_I64("FFH_THIN_ROWS_MIN_BATCH", (int64_t)1 << 40);   // A/B switch
  static const int64_t thin_max = FFH_LAB_I64("FFH_THIN_MAX_BATCH", 16384);
  if (!no_thin && in <= 16 && out >= 64 && out % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)y | (uintptr_t)(bias ? bias : w)) & 15) == 0 && batch >= thin_rows_min) {
    hipLaunchKernelGGL(linear_thin_fwd_rows_kernel, dim3((unsigned)((batch + kThinRowsPerWg - 1) / kThinRowsPerWg), (unsigned)((out + 255) / 256)), dim3(256), 0,
                       as_stream(s), x, ldx, y, ldy, w, bias, in, out, batch, act);
    FFH_LAUNCH_CHECK(c, "linear_thin_fwd_rows_kernel");
    ffh_route_add(c, "linear_fwd|thin_rows");
    return FFH_OK;
  }
  if (!no_thin && in <= 16 && out >= 64 && batch < thin_max) {
    hipLaunchKernelGGL(linear_thin_fwd_kernel, dim3((unsigned)((batch + 31) / 32), (unsigned)((out + 255) / 256)), dim3(512), 0, as_stream(s), x, ldx, y, ldy,
                       w, bias, in, out, batch, act);
    FFH_LAUNCH_CHECK(c, "linear_thin_fwd_kernel");
    ffh_route_add(c, "linear_fwd|thin");
    return FFH_OK;
  }
  GemmArgs g{};
  g.A = x; g.sAm = ldx; g.sAk = 1;
  g.B = w; g.sBn = in; g.sBk = 1;
  g.C = y; g.ldc = ldy; g.bias = bias;
  g.M = (int)batch; g.N = out; g.K = in;
  g.epi = EPI_STORE; g.act = act;
  if (use_bf16(c, in, out, batch)) return launch_gemm_bf16_form(c, g, BF16_FORM_FWD, s, "linear_fwd gemm (bf16)");
  {
    // big aligned layers: the persistent one-workgroup-per-CU kernel (linear_sk.hip)
    const int rc = launch_gemm_sk(c, g, SK_FORM_FWD, s, "linear_fwd gemm");
    if (rc != 0) return rc < 0 ? rc : FFH_OK;
  }
  {
    GldsArgs d{};
    d.A = x; d.lda = ldx; d.B = w; d.ldb = in; d.C = y; d.ldc = ldy; d.bias = bias;
    d.M = (int)batch; d.N = out; d.K = in; d.epi = EPI_STORE; d.act = act;
    const int rc = launch_glds<false, false>(c, d, false, s, "linear_fwd gemm (lds-dma)");
    if (rc != 0) return rc < 0 ? rc : FFH_OK;
  }
  return launch_gemm<true, true>(c, g, 1, s, "linear_fwd gemm");
}

}  // extern "C"

namespace {
// dy <- dy * act'(y) in place (act NONE: dy untouched) and db[o] += sum_b dy[b][o] (db may be null): one pass over dy
int launch_act_bwd_bias(ffh_ctx* c, float* dy, int64_t lddy, const float* y, int64_t ldy, float* db, int out, int64_t batch, int act, ffh_stream st) {
  if (act == FFH_AC_MODE_NONE && !db) return FFH_OK;
  const bool v4 = (out % 4 == 0) && (lddy % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)dy & 15) == 0) && (((uintptr_t)y & 15) == 0);
  const int cols_v = v4 ? out / 4 : out;
  int tx = 1;
  while (tx < cols_v && tx < 256) tx <<= 1;       // power of two: TY = 256 / TX rows in flight per workgroup
  const int ty = 256 / tx;
  int64_t rows = (batch + 2 * c->num_cus - 1) / (2 * c->num_cus);
  if (rows < 4 * ty) rows = 4 * ty;
  if (c->deterministic) rows = batch;             // one workgroup: the column sums meet in a fixed order
  const unsigned grid = (unsigned)((batch + rows - 1) / rows);
  if (v4) hipLaunchKernelGGL((act_bwd_bias_kernel<4>), dim3(grid), dim3(256), 0, as_stream(st), dy, lddy, y, ldy, db, out, batch, (int)rows, act, tx);
  else hipLaunchKernelGGL((act_bwd_bias_kernel<1>), dim3(grid), dim3(256), 0, as_stream(st), dy, lddy, y, ldy, db, out, batch, (int)rows, act, tx);
  FFH_LAUNCH_CHECK(c, "act_bwd_bias_kernel");
  return FFH_OK;
}

// the partial-row scratch of linear_skinny_bwd_kernel for launches on stream s: reserved by ffh_ctx_reserve_scratch(ctx, s) (runtime.hip);
// a compute entry point never allocates.  No set for this stream: the atomic-chain form serves the layer.
bool skinny_ws_for(ffh_ctx* c, hipStream_t s, float** ws, unsigned** cnt) {
  for (int i = 0; i < c->nscratch; i++)
    if (c->scratch[i].stream == (void*)s && c->scratch[i].skinny_ws) { *ws = c->scratch[i].skinny_ws; *cnt = c->scratch[i].skinny_cnt; return true; }
  return false;
}

// label != NULL: the MSE loss step is folded into the one-launch backward (ffh_linear_bwd_mse); layers that path does
// not serve return FFH_ERR_UNSUPPORTED before anything is launched
int linear_bwd_impl(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy,
                    float* dy, int64_t lddy, const float* w, float* dw, float* db,
                    int in, int out, int64_t batch, int act, int flags, ffh_stream s, ffh_stream s_dw,
                    const float* label, float loss_scale, ffh_perf_metrics* perf, int metrics_flags) {
  FFH_REQUIRE(c, in > 0 && out > 0 && batch >= 0 && ldx >= in && ldy >= out && lddy >= out && (!dx || lddx >= in), "linear_bwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x && y && dy && w && dw), "linear_bwd: null pointer");
  FFH_REQUIRE(c, batch < (1LL << 31), "linear_bwd: batch too large");
  if (!act_ok(act)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd: activation not supported (NONE, RELU, SIGMOID; GELU is forward-only, as in the reference)");
  if (batch == 0) return FFH_OK;
  // 1. sigmoid: its gradient is not idempotent, so it gets its own in-place pass (with the bias sums).
  //    relu / none: folded into the GEMMs' operand loads below (no separate pass over dy).
  const bool premasked = (flags & FFH_LINEAR_DY_PREMASKED) != 0;
  if (premasked) act = FFH_AC_MODE_NONE;           // the producer of dy applied the activation derivative already
  const bool mask_by_x = (flags & FFH_LINEAR_DX_MASK_BY_X) != 0;
  const bool separate = act == FFH_AC_MODE_SIGMOID;
  const bool do_dw = !(flags & FFH_LINEAR_ONLY_DX);
  const bool do_dx = !(flags & FFH_LINEAR_ONLY_DW);
  FFH_REQUIRE(c, do_dw || do_dx, "linear_bwd_ex: ONLY_DX and ONLY_DW are exclusive");
  static const int no_skinny = FFH_LAB_INT("FFH_NO_SKINNY", 0);   // A/B switch (tools/ab.sh)
  const bool skinny_vec = !no_skinny && (in % 4 == 0) && glds_aligned(x, ldx) && (((uintptr_t)w & 15) == 0) && (!dx || glds_aligned(dx, lddx));
  // up to 4 outputs with in <= 1024, or up to 16 outputs with in <= 256 (the 64 -> 16 layer in front of the interaction):
  // whole backward in one launch
  const bool skinny_shape = (out <= kSkinnyMaxOut && in <= kSkinnyMaxIn) || (out <= 16 && in <= 256);
  if (skinny_shape && skinny_vec && !c->deterministic) {
    // one launch for the whole layer (the split ONLY_* forms keep their meaning; a forked dw stream is not needed)
    const bool only_dx = !do_dw, only_dw = !do_dx;
    SkinnyBwdArgs a{};
    a.x = x; a.dx = dx; a.y = y; a.dy = dy; a.w = w; a.dw = dw; a.db = db;
    a.ldx = ldx; a.lddx = lddx; a.ldy = ldy; a.lddy = lddy; a.batch = batch; a.in = in; a.out = out;
    a.act = (only_dw && separate) ? FFH_AC_MODE_NONE : act;            // sigmoid: the ONLY_DX call transformed dy already
    a.write_back = a.act != FFH_AC_MODE_NONE && !(only_dx && !separate);  // relu + ONLY_DX: dy is read through the mask, not written
    a.do_db = only_dx ? separate : (only_dw ? !separate : 1);
    a.do_dw = do_dw; a.do_dx = do_dx && dx != nullptr;
    a.dx_overwrite = (flags & FFH_LINEAR_DX_OVERWRITE) ? 1 : 0;
    a.mask_by_x = mask_by_x ? 1 : 0;
    a.dx16 = a.do_dx ? ffh_mirror_of(c, dx, (size_t)((batch - 1) * lddx + in) * 4) : nullptr;      // tensor-op mode: the twin of dx, for the layer below
    if (label) {
      if (only_dx || only_dw || premasked || out > kSkinnyMaxOut) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: not a whole one-launch layer");
      a.label = label; a.loss_scale = loss_scale; a.perf = perf; a.metrics_flags = metrics_flags;
      a.write_back = 1;              // dy must end up holding what the loss step + the layer's in-place pass leave there
    }
    // dW / db: the workgroups' partial rows meet in ctx-owned scratch and the last one to arrive adds them up (no atomic chains); then
    // the number of workgroups follows what the x / dX stream wants.  Without scratch (a capture in flight before the stream's first
    // eager call, more than four streams) they are accumulated with one atomic per weight per workgroup: adds to ONE address serialise
    // (~0.1 us each), so the number of workgroups is kept near 32 -- and every wave then keeps 16 / NC rows in flight to cover the latency
    float* sk_ws = nullptr; unsigned* sk_cnt = nullptr;
    static const int no_ws = FFH_LAB_INT("FFH_SKINNY_NO_WS", 0);     // A/B switch
    const int ws_stride = (out * in + out + 3) / 4 * 4;
    // (from 16384 samples up, where the stream wants 256 workgroups and the chains are 256 long: 256 -> 1 at 32768 samples dW + db
    //  33.4 -> 19.4 us alone.  Below that 32 - 64 workgroups and their short chains are faster than the last arriver's tail:
    //  2048 x 64 -> 16: 11 vs 28 us, Kaggle-shape step 0.186 vs 0.190 ms.  FFH_SKINNY_WS_MIN_BATCH: A/B)
    static const int64_t ws_min_batch = FFH_LAB_I64("FFH_SKINNY_WS_MIN_BATCH", 16384);
    const bool want_ws = !no_ws && batch >= ws_min_batch && (a.do_dw || (a.do_db && db)) && ws_stride <= kSkinnyWsRow && skinny_ws_for(c, as_stream(s), &sk_ws, &sk_cnt);
    static const int nblk_env = FFH_LAB_INT("FFH_SKINNY_NBLK", 0);   // A/B switch
    // large batches: HBM traffic outweighs the longer atomic chains -- at 32768 samples the x / dX traffic (67 MB for the 256 -> 1
    // layer) needs every CU's load queue, and the 256 per-workgroup atomics per weight are ~3 us spread over the launch
    const int64_t nblk = nblk_env > 0 ? nblk_env : (want_ws ? (batch >= 16384 ? 256 : 128) : (batch >= 16384 ? 256 : (batch >= 8192 ? 64 : 32)));
    int64_t rpb = (batch + nblk - 1) / nblk;
    rpb = (rpb + 15) / 16 * 16;
    if (rpb > 1024) rpb = 1024;
    a.rows_per_block = (int)rpb;
    const unsigned grid = (unsigned)((batch + rpb - 1) / rpb);
    size_t lds = (size_t)rpb * out * sizeof(float);
    const size_t red = (size_t)4 * out * in * sizeof(float);            // cross-wave dW reduction
    if (red > lds) lds = red;
    if (want_ws && grid <= (unsigned)kSkinnyWsBlocks) {
      a.ws = sk_ws; a.ws_cnt = sk_cnt; a.ws_stride = ws_stride;
      if (lds < 4096) lds = 4096;                          // the last arriver's cross-thread combine
    }
    const int nc = in <= 256 ? 1 : (in <= 512 ? 2 : 4);
#define FFH_SKINNY(NCV, NOV) hipLaunchKernelGGL((linear_skinny_bwd_kernel<NCV, NOV>), dim3(grid), dim3(256), lds, as_stream(s), a)
    if (out == 1) { if (nc == 1) FFH_SKINNY(1, 1); else if (nc == 2) FFH_SKINNY(2, 1); else FFH_SKINNY(4, 1); }
    else if (out <= 4) { if (nc == 1) FFH_SKINNY(1, 4); else if (nc == 2) FFH_SKINNY(2, 4); else FFH_SKINNY(4, 4); }
    else if (in == 64) hipLaunchKernelGGL((linear_skinny_bwd_kernel<1, 16, 4>), dim3(grid), dim3(256), lds, as_stream(s), a);
    else if (in == 128) hipLaunchKernelGGL((linear_skinny_bwd_kernel<1, 16, 2>), dim3(grid), dim3(256), lds, as_stream(s), a);
    else FFH_SKINNY(1, 16);
#undef FFH_SKINNY
    FFH_LAUNCH_CHECK(c, "linear_skinny_bwd_kernel");
    ffh_route_add(c, "linear_bwd|skinny");
    if (a.do_dx) {      // split mode: the three-plane image of dx (the dy of the layer below), by a pass over what the launch stored
      int col0 = 0;
      if (ffh_planes_of(c, dx, (size_t)((batch - 1) * lddx + in) * 4, &col0)) return ffh_convert_f32_to_bf16x3(c, dx, batch, in, lddx, s);
    }
    return FFH_OK;
  }
  if (label) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: not a one-launch layer (out_dim <= 4, in_dim <= 1024, aligned)");
  if (use_bf16(c, in, out, batch)) {
    // tensor-op math mode: the activation gradient (and db) as its own fp32 pass over dy, then the two GEMMs on bf16 operands
    bool relu_ = act == FFH_AC_MODE_RELU;
    auto act_pass = [&](ffh_stream st, int a) -> int { return launch_act_bwd_bias(c, dy, lddy, y, ldy, db, out, batch, a, st); };
    if (separate && do_dx) { const int rc = act_pass(s, act); if (rc) return rc; }       // sigmoid: in place, with db, before any fork
    // A live relu' with both gradients wanted (round 5): the pass over dy FIRST, on s, in front of the fork -- 12 us at 32768 x 128 -- and then the
    // two GEMMs as for a final dy.  Before, the pass ran on the weight-gradient stream while the data-gradient GEMM read dy through
    // relu'(y) (the masking 128 x 128 kernel: y as a third operand, the mask applied in registers); that kernel heads the mode's critical
    // chain in the DLRM step (the bottom MLP's last layer: 301 us beside the table update).  Same values, same kernels' arithmetic.
    // Measured: the call alone 134 -> 114 us at 32768 x 256 -> 128; the step level (1.95-2.01 ms either way: that stretch is bound by what runs beside it).
    static const int relu_first = FFH_LAB_INT("FFH_BF16_RELU_FIRST", 1);          // A/B switch
    bool relu_done = false;
    if (relu_first && relu_ && !separate && do_dw && do_dx && dx) {
      const int rc = act_pass(s, act);
      if (rc) return rc;
      relu_ = false; relu_done = true;
    }
    const bool forked_ = do_dw && do_dx && s_dw != nullptr && s_dw != s;
    ffh_stream sw_ = forked_ ? s_dw : s;
    if (forked_) {
      c->second_stream_used = 1;
      if (!c->ev_fork) FFH_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      FFH_HIP_TRY(c, hipEventRecord(c->ev_fork, as_stream(s)));
      FFH_HIP_TRY(c, hipStreamWaitEvent(as_stream(sw_), c->ev_fork, 0));
    }
    if (do_dw) {
      // no activation gradient to apply (premasked dy / no activation): the pass would only sum db -- the LDS-DMA weight-gradient
      // kernel does that itself from the fp32 dy where it serves the layer
      const bool db_only = !separate && !relu_done && act == FFH_AC_MODE_NONE && db != nullptr;
      if (!separate && !db_only && !relu_done) { const int rc = act_pass(sw_, act); if (rc) return rc; }          // relu: mask written back in place (idempotent); db
      GemmArgs gw{};
      gw.A = dy; gw.sAm = 1; gw.sAk = lddy;
      gw.B = x; gw.sBn = 1; gw.sBk = ldx;
      gw.C = dw; gw.ldc = in;
      gw.M = out; gw.N = in; gw.K = (int)batch;
      gw.epi = EPI_ATOMIC; gw.act = FFH_AC_MODE_NONE;
      gw.a_not_twinned = act != FFH_AC_MODE_NONE;       // a live activation gradient rewrote dy in place: its bf16 twin is stale
      if (db_only) gw.db = db;
      const int rc = launch_gemm_bf16_form(c, gw, BF16_FORM_DW, sw_, "linear_bwd dw gemm (bf16)");
      if (rc) return rc;
      if (db_only && !gw.db_done) { const int rc2 = act_pass(sw_, act); if (rc2) return rc2; }
    }
    if (dx && do_dx) {
      GemmArgs gx{};
      gx.A = dy; gx.sAm = lddy; gx.sAk = 1;
      gx.B = w; gx.sBn = 1; gx.sBk = in;
      gx.C = dx; gx.ldc = lddx;
      gx.M = (int)batch; gx.N = in; gx.K = out;
      gx.epi = (flags & FFH_LINEAR_DX_OVERWRITE) ? EPI_STORE : EPI_ADD;
      gx.act = FFH_AC_MODE_NONE;
      gx.a_not_twinned = act != FFH_AC_MODE_NONE;
      if (mask_by_x) { gx.mask = x; gx.ldmask = ldx; }
      int rc;
      if ((forked_ || !do_dw) && relu_) {          // dy is (being) masked by someone else: read it through relu'(y)
        gx.act_y = y; gx.ld_act_y = ldy;
        rc = launch_gemm_bf16_form(c, gx, BF16_FORM_DX_MASK, s, "linear_bwd dx gemm (bf16, masking)");
      } else {
        rc = launch_gemm_bf16_form(c, gx, BF16_FORM_DX, s, "linear_bwd dx gemm (bf16)");
      }
      if (rc) return rc;
    }
    return FFH_OK;
  }
  if (separate && do_dx) {
    const int rc = launch_act_bwd_bias(c, dy, lddy, y, ldy, db, out, batch, act, s);
    if (rc) return rc;
  }
  // big aligned layers: the persistent one-workgroup-per-CU kernels (linear_sk.hip).  They take dy as it is, so a live relu'
  // (not premasked) and the bias gradient go through one pass over dy first -- on s, in front of the fork, so that both GEMMs
  // read the finished dy.  Not served: ONLY_DX with a live relu (dy must not be written there), the column-map epilogue.
  {
    const bool want_dx = dx && do_dx;
    const bool relu_live = act == FFH_AC_MODE_RELU;
    const bool scatter_pending = c->scatter_map && c->scatter_ncols == in && (flags & FFH_LINEAR_DX_OVERWRITE);
    GemmArgs gw{}, gx{};
    gw.A = dy; gw.sAm = 1; gw.sAk = lddy; gw.B = x; gw.sBn = 1; gw.sBk = ldx; gw.C = dw; gw.ldc = in;
    gw.M = out; gw.N = in; gw.K = (int)batch; gw.epi = EPI_ATOMIC; gw.act = FFH_AC_MODE_NONE;
    gx.A = dy; gx.sAm = lddy; gx.sAk = 1; gx.B = w; gx.sBn = 1; gx.sBk = in; gx.C = dx; gx.ldc = lddx;
    gx.M = (int)batch; gx.N = in; gx.K = out; gx.epi = (flags & FFH_LINEAR_DX_OVERWRITE) ? EPI_STORE : EPI_ADD; gx.act = FFH_AC_MODE_NONE;
    if (mask_by_x) { gx.mask = x; gx.ldmask = ldx; }
    // ffh_linear_bwd_set_dx_scatter: the persistent data-gradient kernel takes the column map in its epilogue (SK_EPI_DX_CMAP)
    const bool scatter_sk = want_dx && scatter_pending && !c->deterministic;
    if (scatter_sk) gx.colmap = (const ffh_col_dest*)c->scatter_map;
    // ffh_linear_bwd_set_dx_colsum: the lower layer's bias gradient from this kernel's store epilogue
    const bool colsum_sk = want_dx && c->colsum_dst && c->colsum_ncols == in && gx.epi == EPI_STORE && !scatter_pending && !c->deterministic;
    if (colsum_sk) gx.colsum = c->colsum_dst;
    bool ok = (do_dw || want_dx) && !(relu_live && !do_dw) && !(want_dx && scatter_pending && !scatter_sk);
    // A layer whose data gradient is a shape of the persistent kernel while its weight gradient is not (few k-tiles per workgroup:
    // 8192 x 512 -> 256) used to lose BOTH to the other kernels (dX 51 us in the MLPerf-shape step instead of 23).  With dy final
    // (no live activation derivative) the call is the two split calls it stands for: ONLY_DX here, ONLY_DW through whatever serves it,
    // forked to s_dw as the caller asked.
    if (ok && do_dw && want_dx && !relu_live && !separate && !label && !scatter_pending && !gemm_sk_serves(c, gw, SK_FORM_DW) && gemm_sk_serves(c, gx, SK_FORM_DX)) {
      static const int dw_on_s = FFH_LAB_INT("FFH_SPLIT_DW_ON_S", 0);      // A/B: the non-persistent dW behind the dX on s instead of beside persistent kernels on s_dw
      const bool forked_split = s_dw != nullptr && s_dw != s && !dw_on_s;
      if (forked_split) {
        c->second_stream_used = 1;
        if (!c->ev_fork) FFH_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        FFH_HIP_TRY(c, hipEventRecord(c->ev_fork, as_stream(s)));
        FFH_HIP_TRY(c, hipStreamWaitEvent(as_stream(s_dw), c->ev_fork, 0));
      }
      const int f0 = flags & ~(FFH_LINEAR_ONLY_DX | FFH_LINEAR_ONLY_DW);
      int rc = linear_bwd_impl(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, batch, act, f0 | FFH_LINEAR_ONLY_DX, s, nullptr, nullptr, 0.0f, nullptr, 0);
      if (rc) return rc;
      rc = linear_bwd_impl(c, x, ldx, nullptr, lddx, y, ldy, dy, lddy, w, dw, db, in, out, batch, act, f0 | FFH_LINEAR_ONLY_DW, forked_split ? s_dw : s, nullptr, nullptr, 0.0f, nullptr, 0);
      return rc;
    }
    if (ok && do_dw) ok = gemm_sk_serves(c, gw, SK_FORM_DW);
    if (ok && want_dx) ok = gemm_sk_serves(c, gx, SK_FORM_DX);
    if (ok) {
      const bool forked_sk = do_dw && want_dx && s_dw != nullptr && s_dw != s;
      ffh_stream sw_sk = forked_sk ? s_dw : s;
      if (do_dw && relu_live) {      // live relu': masked in place, with the bias gradient, in one pass in front of both GEMMs
        const int rc = launch_act_bwd_bias(c, dy, lddy, y, ldy, db, out, batch, FFH_AC_MODE_RELU, s);
        if (rc) return rc;
      } else if (do_dw && !separate) {
        gw.db = db;                  // dy is final already (none / premasked): its column sums ride on the weight-gradient kernel
      }                              // (sigmoid: its own pass above, or the ONLY_DX call before this one, has produced db)
      if (forked_sk) {
        c->second_stream_used = 1;
        if (!c->ev_fork) FFH_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        FFH_HIP_TRY(c, hipEventRecord(c->ev_fork, as_stream(s)));
        FFH_HIP_TRY(c, hipStreamWaitEvent(as_stream(sw_sk), c->ev_fork, 0));
      }
      // the data gradient first: it is what the layer below waits for; the weight gradient fills the chip behind it
      if (want_dx) {
        const int rc = launch_gemm_sk(c, gx, SK_FORM_DX, s, "linear_bwd dx gemm");
        if (rc < 0) return rc;
        if (rc == 1 && colsum_sk) c->colsum_used = 1;
        if (rc == 1 && scatter_sk) {
          c->scatter_used = 1;
          if (c->scatter_event) FFH_HIP_TRY(c, hipEventRecord((hipEvent_t)c->scatter_event, as_stream(s)));   // "gradients ready" behind the kernel that produced them
        }
        if (rc == 0) {     // (the launch was refused after the plan said yes: the register-staged kernel takes the data gradient, map included)
          GemmArgs g2 = gx;
          const int rc2 = scatter_sk ? launch_gemm<true, false, false, true>(c, g2, 1, s, "linear_bwd dx gemm (column map)") : launch_gemm<true, false>(c, g2, 1, s, "linear_bwd dx gemm");
          if (rc2) return rc2;
          if (scatter_sk) { c->scatter_used = 1; if (c->scatter_event) FFH_HIP_TRY(c, hipEventRecord((hipEvent_t)c->scatter_event, as_stream(s))); }
        }
      }
      if (do_dw) {
        const int rc = launch_gemm_sk(c, gw, SK_FORM_DW, sw_sk, "linear_bwd dw gemm");
        if (rc < 0) return rc;
        if (rc == 0) { GemmArgs g2 = gw; g2.act_y = y; g2.ld_act_y = ldy; g2.fuse = g2.db ? 2 : 0; const int rc2 = launch_gemm<false, false, true>(c, g2, 1, sw_sk, "linear_bwd dw gemm"); if (rc2) return rc2; }
      }
      return FFH_OK;
    }
  }
  // mid-size layer with nothing to mask while loading: data- and weight-gradient GEMMs in ONE LDS-DMA launch on s
  // (measured against the two-stream form: fewer barrier packets on the critical stream, one dispatch)
  if (do_dw && do_dx && dx && act != FFH_AC_MODE_RELU) {
    GldsArgs dxg{}, dwg{};
    dxg.A = dy; dxg.lda = lddy; dxg.B = w; dxg.ldb = in; dxg.C = dx; dxg.ldc = lddx;
    dxg.M = (int)batch; dxg.N = in; dxg.K = out; dxg.epi = (flags & FFH_LINEAR_DX_OVERWRITE) ? EPI_STORE : EPI_ADD; dxg.act = FFH_AC_MODE_NONE;
    if (mask_by_x) { dxg.mask = x; dxg.ldmask = ldx; }
    dwg.A = dy; dwg.lda = lddy; dwg.B = x; dwg.ldb = ldx; dwg.C = dw; dwg.ldc = in;
    dwg.M = out; dwg.N = in; dwg.K = (int)batch; dwg.epi = EPI_ATOMIC; dwg.act = FFH_AC_MODE_NONE;
    dwg.db = separate ? nullptr : db;
    const int rc = launch_glds_bwd(c, dxg, dwg, s);
    if (rc < 0) return rc;
    if (rc == 1) return FFH_OK;
  }
  // the weight-gradient GEMM may go to its own stream: it only needs dy (and y, x), which are ready on s now
  const bool forked = do_dw && do_dx && s_dw != nullptr && s_dw != s;
  ffh_stream sw = forked ? s_dw : s;
  if (forked) {
    c->second_stream_used = 1;
    if (!c->ev_fork) FFH_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    FFH_HIP_TRY(c, hipEventRecord(c->ev_fork, as_stream(s)));
    FFH_HIP_TRY(c, hipStreamWaitEvent(as_stream(sw), c->ev_fork, 0));
  }
  const bool relu = act == FFH_AC_MODE_RELU;
  // 2. dw[o][i] += sum_b dy[b][o] x[b][i]   (split-K over the batch, fp32 atomics); relu' mask applied on load and
  //    written back to dy in place, db = column sums of the same tiles
  if (do_dw) {
    GemmArgs g{};
    g.A = dy; g.sAm = 1; g.sAk = lddy;
    g.B = x; g.sBn = 1; g.sBk = ldx;
    g.C = dw; g.ldc = in;
    g.M = out; g.N = in; g.K = (int)batch;
    g.epi = EPI_ATOMIC; g.act = FFH_AC_MODE_NONE;
    g.act_y = y; g.ld_act_y = ldy; g.db = db;
    g.fuse = separate ? 0 : ((relu ? 1 : 0) | (db ? 2 : 0));
    int rc = 0;
    if (!relu) {
      // no mask to apply while loading: eligible for the LDS-DMA kernel (db from its LDS image)
      GldsArgs d{};
      d.A = dy; d.lda = lddy; d.B = x; d.ldb = ldx; d.C = dw; d.ldc = in;
      d.M = out; d.N = in; d.K = (int)batch; d.epi = EPI_ATOMIC; d.act = FFH_AC_MODE_NONE;
      d.db = separate ? nullptr : db;
      rc = launch_glds<true, true>(c, d, true, sw, "linear_bwd dw gemm (lds-dma)");
      if (rc < 0) return rc;
    }
    if (rc == 0) rc = launch_gemm<false, false, true>(c, g, 1, sw, "linear_bwd dw gemm");
    else rc = 0;
    if (rc) return rc;
  }
  // 3. dx[b][i] (+)= sum_o dy[b][o] w[o][i].  Unforked it runs behind the dw GEMM and reads the masked dy;
  //    forked it masks dy itself while loading (idempotent, so the concurrent in-place write-back is harmless)
  if (dx && do_dx) {
    GemmArgs g{};
    g.A = dy; g.sAm = lddy; g.sAk = 1;
    g.B = w; g.sBn = 1; g.sBk = in;
    g.C = dx; g.ldc = lddx;
    g.M = (int)batch; g.N = in; g.K = out;
    g.epi = (flags & FFH_LINEAR_DX_OVERWRITE) ? EPI_STORE : EPI_ADD;
    g.act = FFH_AC_MODE_NONE;
    if (mask_by_x) { g.mask = x; g.ldmask = ldx; }
    int rc;
    if (!relu) {
      GldsArgs d{};
      d.A = dy; d.lda = lddy; d.B = w; d.ldb = in; d.C = dx; d.ldc = lddx;
      d.M = (int)batch; d.N = in; d.K = out; d.epi = g.epi; d.act = FFH_AC_MODE_NONE;
      if (mask_by_x) { d.mask = x; d.ldmask = ldx; }
      rc = launch_glds<false, true>(c, d, false, s, "linear_bwd dx gemm (lds-dma)");
      if (rc < 0) return rc;
      if (rc == 1) return FFH_OK;
    }
    // ffh_linear_bwd_set_dx_scatter: the data gradient goes where a Concat backward would copy it (the register-staged kernel's
    // epilogue takes the column map too: no pack kernel on the critical stream in front of the all-to-all)
    const bool scatter = c->scatter_map && c->scatter_ncols == in && g.epi == EPI_STORE && !c->deterministic;
    if (scatter) g.colmap = (const ffh_col_dest*)c->scatter_map;
    if ((forked || !do_dw) && relu) {
      g.act_y = y; g.ld_act_y = ldy; g.fuse = 1;
      rc = scatter ? launch_gemm<true, false, true, true>(c, g, 1, s, "linear_bwd dx gemm (masking, column map)")
                   : launch_gemm<true, false, true>(c, g, 1, s, "linear_bwd dx gemm (masking)");
    } else {
      rc = scatter ? launch_gemm<true, false, false, true>(c, g, 1, s, "linear_bwd dx gemm (column map)")
                   : launch_gemm<true, false>(c, g, 1, s, "linear_bwd dx gemm");
    }
    if (rc == FFH_OK && scatter) {
      c->scatter_used = 1;
      if (c->scatter_event) {          // "gradients ready" behind the kernel that produced them
        FFH_HIP_TRY(c, hipEventRecord((hipEvent_t)c->scatter_event, as_stream(s)));
      }
    }
    if (rc) return rc;
  }
  return FFH_OK;
}
}  // namespace

extern "C" {

int ffh_linear_bwd_ex(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy,
                      float* dy, int64_t lddy, const float* w, float* dw, float* db,
                      int in, int out, int64_t batch, int act, int flags, ffh_stream s, ffh_stream s_dw) {
  if (c) { c->scatter_used = 0; c->colsum_used = 0; }
  ffh_route_clear(c);
  const int rc = linear_bwd_impl(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, batch, act, flags, s, s_dw, nullptr, 0.0f, nullptr, 0);
  if (c) { c->scatter_map = nullptr; c->scatter_event = nullptr; c->colsum_dst = nullptr; }       // one call only, taken or not
  if (c && c->attach_event) {        // no launch could carry it: the ordinary record behind everything this call put on s
    hipEvent_t ev = (hipEvent_t)c->attach_event;
    c->attach_event = nullptr;
    if (rc == FFH_OK) FFH_HIP_TRY(c, hipEventRecord(ev, as_stream(s)));
  }
  return rc;
}

int ffh_linear_bwd_set_dx_scatter(ffh_ctx* c, const ffh_col_dest* map, int ncols, ffh_event attach_if_used) {
  if (!c || !map || ncols <= 0) return FFH_ERR_BAD_ARG;
  c->scatter_map = map; c->scatter_ncols = ncols; c->scatter_event = attach_if_used; c->scatter_used = 0;
  return FFH_OK;
}
int ffh_linear_dx_scatter_used(ffh_ctx* c) { return c ? c->scatter_used : 0; }

int ffh_linear_bwd_set_dx_colsum(ffh_ctx* c, float* colsum, int ncols) {
  if (!c || !colsum || ncols <= 0) return FFH_ERR_BAD_ARG;
  c->colsum_dst = colsum; c->colsum_ncols = ncols; c->colsum_used = 0;
  return FFH_OK;
}
int ffh_linear_dx_colsum_used(ffh_ctx* c) { return c ? c->colsum_used : 0; }

int ffh_event_record_with_next_linear_bwd(ffh_ctx* c, ffh_event e) {
  if (!c || !e) return FFH_ERR_BAD_ARG;
  c->attach_event = e;
  return FFH_OK;
}

int ffh_linear_bwd_mse(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy,
                       float* dy, int64_t lddy, const float* w, float* dw, float* db,
                       int in, int out, int64_t batch, int act, int flags,
                       const float* label, float scale, ffh_perf_metrics* perf, int metrics_flags, ffh_stream s) {
  FFH_REQUIRE(c, label && perf, "linear_bwd_mse: label and perf are required");
  if (flags & (FFH_LINEAR_ONLY_DX | FFH_LINEAR_ONLY_DW | FFH_LINEAR_DY_PREMASKED)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: split / premasked forms");
  if (ldy != out || lddy != out) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd_mse: y and dy must be contiguous [batch][out_dim]");
  if (batch == 0) return FFH_OK;
  ffh_route_clear(c);
  c->colsum_used = 0;
  const int rc = linear_bwd_impl(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, batch, act, flags, s, nullptr, label, scale, perf, metrics_flags);
  c->colsum_dst = nullptr;             // ffh_linear_bwd_set_dx_colsum: one call only, taken or not (this launch never takes it)
  return rc;
}

int ffh_linear_pair_fwd(ffh_ctx* c, const float* x_l, int64_t ldx_l, const float* w_l, const float* b_l, int in_l, int act_l,
                        float* y_l, int64_t ldy_l, int mid, const float* w_u, const float* b_u, int out_u, int act_u,
                        float* y_u, int64_t ldy_u, int64_t batch, ffh_stream s) {
  FFH_REQUIRE(c, in_l > 0 && mid > 0 && out_u > 0 && batch >= 0 && ldx_l >= in_l && ldy_l >= mid && ldy_u >= out_u, "linear_pair_fwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x_l && w_l && y_l && w_u && y_u), "linear_pair_fwd: null pointer");
  if (!act_ok(act_l) || !act_ok(act_u)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_fwd: activation not supported (NONE, RELU, SIGMOID)");
  const int ks = (mid == 32 || mid == 64) ? 8 / (mid / 32) : 1;
  if (out_u > 16 || (mid != 32 && mid != 64) || in_l % (32 * ks) != 0 || batch >= (1LL << 31) || ldx_l % 4 != 0 ||
      (((uintptr_t)x_l | (uintptr_t)w_l | (uintptr_t)w_u) & 15) != 0)
    return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_fwd: shapes (out_u <= 16, mid 32 or 64, in_l a multiple of 32 k-slices, 16-byte aligned operands)");
  if (batch == 0) return FFH_OK;
  PairFwdArgs a{};
  a.xl = x_l; a.ldxl = ldx_l; a.wl = w_l; a.bl = b_l; a.in_l = in_l; a.act_l = act_l; a.yl = y_l; a.ldyl = ldy_l; a.mid = mid;
  a.wu = w_u; a.bu = b_u; a.out_u = out_u; a.act_u = act_u; a.yu = y_u; a.ldyu = ldy_u; a.batch = batch;
  hipLaunchKernelGGL(linear_pair_fwd_kernel, dim3((unsigned)((batch + 31) / 32)), dim3(512), 0, as_stream(s), a);
  FFH_LAUNCH_CHECK(c, "linear_pair_fwd_kernel");
  return FFH_OK;
}

int ffh_linear_pair_bwd(ffh_ctx* c, const float* x_u, int64_t ldx_u, const float* y_u, int64_t ldy_u, float* dy_u, int64_t lddy_u,
                        const float* w_u, float* dw_u, float* db_u, int in_u, int out_u, int act_u, int flags_u,
                        const float* x_l, int64_t ldx_l, float* dx_l, int64_t lddx_l, float* dy_l, int64_t lddy_l, const float* w_l,
                        int in_l, int act_l, int flags_l, int64_t batch, ffh_stream s) {
  FFH_REQUIRE(c, in_u > 0 && out_u > 0 && in_l > 0 && batch >= 0 && ldx_u >= in_u && ldy_u >= out_u && lddy_u >= out_u && ldx_l >= in_l &&
                     lddx_l >= in_l && lddy_l >= in_u, "linear_pair_bwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x_u && y_u && dy_u && w_u && dw_u && x_l && dx_l && dy_l && w_l), "linear_pair_bwd: null pointer");
  if (!act_ok(act_u)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: activation not supported (NONE, RELU, SIGMOID)");
  if ((flags_u & ~FFH_LINEAR_DY_PREMASKED) || (flags_l & ~(FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X)))
    return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: flags");
  if (c->deterministic) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: its dW / db partials meet by atomics (deterministic mode)");
  if (out_u > 16 || (in_u != 32 && in_u != 64) || in_l % 32 != 0 || (act_l != FFH_AC_MODE_RELU && act_l != FFH_AC_MODE_NONE) || batch >= (1LL << 31))
    return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_pair_bwd: shapes (out_u <= 16, in_u 32 or 64, in_l a multiple of 32; lower activation relu / none)");
  if (batch == 0) return FFH_OK;
  PairBwdArgs a{};
  a.xu = x_u; a.ldxu = ldx_u; a.dyu = dy_u; a.lddyu = lddy_u; a.yu = y_u; a.ldyu = ldy_u; a.wu = w_u; a.dwu = dw_u; a.dbu = db_u;
  a.in_u = in_u; a.out_u = out_u; a.act_u = (flags_u & FFH_LINEAR_DY_PREMASKED) ? FFH_AC_MODE_NONE : act_u;
  a.xl = x_l; a.ldxl = ldx_l; a.dxl = dx_l; a.lddxl = lddx_l; a.dyl = dy_l; a.lddyl = lddy_l; a.wl = w_l;
  a.in_l = in_l; a.act_l = act_l; a.dxl_overwrite = (flags_l & FFH_LINEAR_DX_OVERWRITE) ? 1 : 0; a.dxl_mask_by_x = (flags_l & FFH_LINEAR_DX_MASK_BY_X) ? 1 : 0;
  a.batch = batch;
  hipLaunchKernelGGL(linear_pair_bwd_kernel, dim3((unsigned)((batch + 31) / 32)), dim3(512), 0, as_stream(s), a);
  FFH_LAUNCH_CHECK(c, "linear_pair_bwd_kernel");
  return FFH_OK;
}

int ffh_second_stream_used(ffh_ctx* c, int clear) {
  if (!c) return 0;
  const int v = c->second_stream_used;
  if (clear) c->second_stream_used = 0;
  return v;
}

int ffh_linear_bwd(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy,
                   float* dy, int64_t lddy, const float* w, float* dw, float* db,
                   int in, int out, int64_t batch, int act, ffh_stream s) {
  return ffh_linear_bwd_ex(c, x, ldx, dx, lddx, y, ldy, dy, lddy, w, dw, db, in, out, batch, act, 0, s, nullptr);
}

int ffh_bmm_fwd(ffh_ctx* c, float* o, const float* a, const float* b, int m, int n, int k, int64_t batch,
                int asd, int bsd, int seq, ffh_stream s) {
  FFH_REQUIRE(c, m > 0 && n > 0 && k > 0 && batch >= 0, "bmm_fwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (o && a && b), "bmm_fwd: null pointer");
  // strides from the full sizes, then seq_length truncation [ref: src/ops/batch_matmul.cu:212-236]
  const int lda = k, ldb = m, ldo = m;
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
  if (asd == 0 && seq >= 0) { FFH_REQUIRE(c, seq <= k && bsd == 1, "bmm_fwd: seq_length"); k = seq; }
  else if (asd == 1 && seq >= 0) { FFH_REQUIRE(c, seq <= n, "bmm_fwd: seq_length"); n = seq; }
  else FFH_REQUIRE(c, asd < 0 || seq < 0, "bmm_fwd: a_seq_length_dim");
  if (bsd == 0 && seq >= 0) { FFH_REQUIRE(c, seq <= m, "bmm_fwd: seq_length"); m = seq; }
  else if (bsd == 1 && seq >= 0) { FFH_REQUIRE(c, asd == 0 && k == seq, "bmm_fwd: seq_length"); }
  else FFH_REQUIRE(c, bsd < 0 || seq < 0, "bmm_fwd: b_seq_length_dim");
  if (batch == 0 || m == 0 || n == 0 || k == 0) return FFH_OK;
  GemmArgs g{};
  g.A = a; g.sAm = lda; g.sAk = 1; g.bsA = sa;
  g.B = b; g.sBn = 1; g.sBk = ldb; g.bsB = sb;
  g.C = o; g.ldc = ldo; g.bsC = so;
  g.M = n; g.N = m; g.K = k;
  g.epi = EPI_STORE; g.act = FFH_AC_MODE_NONE;
  return launch_gemm<true, false>(c, g, batch, s, "bmm_fwd gemm");
}

int ffh_bmm_bwd(ffh_ctx* c, const float* og, const float* a, float* ag, const float* b, float* bg,
                int m, int n, int k, int64_t batch, ffh_stream s) {
  FFH_REQUIRE(c, m > 0 && n > 0 && k > 0 && batch >= 0, "bmm_bwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (og && a && ag && b && bg), "bmm_bwd: null pointer");
  if (batch == 0) return FFH_OK;
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
  {  // a_grad[r][q] += sum_c og[r][c] * b[q][c]
    GemmArgs g{};
    g.A = og; g.sAm = m; g.sAk = 1; g.bsA = so;
    g.B = b; g.sBn = m; g.sBk = 1; g.bsB = sb;
    g.C = ag; g.ldc = k; g.bsC = sa;
    g.M = n; g.N = k; g.K = m;
    g.epi = EPI_ADD; g.act = FFH_AC_MODE_NONE;
    int rc = launch_gemm<true, true>(c, g, batch, s, "bmm_bwd a_grad gemm");
    if (rc) return rc;
  }
  {  // b_grad[q][cc] += sum_r a[r][q] * og[r][cc]
    GemmArgs g{};
    g.A = a; g.sAm = 1; g.sAk = k; g.bsA = sa;
    g.B = og; g.sBn = 1; g.sBk = m; g.bsB = so;
    g.C = bg; g.ldc = m; g.bsC = sb;
    g.M = k; g.N = m; g.K = n;
    g.epi = EPI_ADD; g.act = FFH_AC_MODE_NONE;
    int rc = launch_gemm<false, false>(c, g, batch, s, "bmm_bwd b_grad gemm");
    if (rc) return rc;
  }
  return FFH_OK;
}

}  // extern "C"
