// linear_bf16_dma.hip -- tensor-op math mode (bf16 operands, fp32 accumulate), the big layers: operands come from their bf16
// twins (ffh_ctx_bf16_mirror_set) straight into LDS by LDS-DMA, two wave groups take turns on the matrix pipe.
//
// 256 x 256 x 64 tiles, one workgroup of 8 waves per CU (two per SIMD), v_mfma_f32_16x16x32_bf16, two 64 KB LDS buffers filled by
// buffer_load_dwordx4 ... lds (no staging registers, no ds_write pass).
//   * The waves are 2 (row groups) x 4 (column groups); a wave owns rows {64g..64g+63} + {128+64g..} and columns {32c..32c+31} +
//     {128+32c..} of the tile: 8 x 4 accumulators of 16 x 16.  A k-tile is four PHASES of 16 MFMAs (one quadrant of the wave's
//     output x 64 k):  1: A-lo x B-lo   2: A-lo x B-hi   3: A-hi x B-hi   4: A-hi x B-lo, each
//         { the phase's fragment reads (4 B + 8 A / 4 B / 8 A / none);  2 DMA pieces of a later k-tile;  s_waitcnt vmcnt(8);
//           s_barrier;  16 MFMAs at raised priority;  s_barrier }
//   * The two row groups run HALF A PHASE APART (group 1 passes one extra barrier at the start): while one wave of a SIMD issues
//     its 16 MFMAs the other issues its reads and DMA pieces and waits at the barrier -- the matrix pipe always has work, and
//     nothing in the loop needs hand placement.
//   * An operand's k-tile is two UNITS of 16 KB (lo: rows / columns 0..127 of the tile, hi: 128..255), each read in exactly one
//     phase (A-lo, B-lo in 1, B-hi in 2, A-hi in 3) and restaged two phases after that read at the earliest:
//         phase 1 stages B-hi(t+1), 2: A-hi(t+1), 3: B-lo(t+2), 4: A-lo(t+2);
//     every phase waits vmcnt(8) after issuing its two pieces: four units stay in flight, a unit is waited for four phases after its
//     issue and one phase -- with a barrier of both groups in between -- before its first read (an LDS-DMA is ordered for a
//     ds_read only by the issuing wave's vmcnt wait followed by a barrier the reader has passed).
//   * LDS images are lane-linear (the DMA writes base + 16 * lane); the swizzles live on the SOURCE address and in the reads:
//       k-contiguous operand: 128 unit-rows x 128 B, 16-byte chunk j of row u at slot j ^ ((u >> 1) & 7): a fragment (16 rows x 32 k)
//         is one conflict-free ds_read_b128 per lane;
//       rows-are-k operand: 64 k-rows x 256 B, chunk j (8 columns) of k-row r at slot j ^ (((r & 3) << 2) | ((r >> 2) & 3)); a
//         fragment is two ds_read_b64_tr_b16 (4 k x 16 columns each, transposed on the way out).
//   * Rows / columns beyond the matrix: the buffer descriptor ends with the operand, so loads past it return 0 and touch
//     nothing; columns past N of a rows-are-k operand read the next row's values into accumulators nobody stores.
//   * Epilogue through LDS (all of it is free by then: 20 KB per wave, half of the wave's accumulators at a time), so that a store
//     covers 8 rows x 128 contiguous bytes and an atomic instruction 2 rows x 128: bias + activation (forward), relu'(x) mask
//     and store / add (dX), atomics onto the k-slices' common tile (dW); the fp32 result and, where the output has a twin, its
//     bf16 rounding.  (A first version went block by block through 4 KB per wave -- 32 dependent LDS round trips and 32
//     dependent bias loads: 13 us per tile where this one takes ~6.)
// The main loop runs at ~1.4 PFLOP/s (32768 x 3456 x 1024); what a launch adds to that is its output traffic (fp32 + bf16:
// 6 bytes per element at the HBM write rate), which this structure does not overlap with the MFMAs.  Developed in
// tools/lab/gemm_bf16_lab.hip (stand-alone, with ablation modes).
//
// Arithmetic: products of bf16 operands are exact in fp32; sums in fp32 in this kernel's own order (k in steps of 32 through
// the MFMA's adder tree) -- the same contract as linear_bf16.hip, compared with the oracle's same-mode result at 1e-5 of the
// term mass.  Replaces cublasSgemm under CUBLAS_TENSOR_OP_MATH [ref: src/runtime/model.cu:81-83; src/ops/linear.cu:436-453,624-659].
#include "linear_gemm.h"

#include <stdlib.h>

using namespace ffh_gemm;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int DM_BM = 256, DM_BN = 256, DM_BK = 64;
constexpr int DM_UNIT = 16384, DM_BUF = 4 * DM_UNIT;
constexpr int DM_LDS = 2 * DM_BUF + 32768;             // two buffers + 32 KB (bias-gradient reduction); the epilogue reuses all 160 KB
constexpr int DM_ALO = 0, DM_AHI = 1, DM_BLO = 2, DM_BHI = 3;

enum { DM_EPI_FWD = 0, DM_EPI_DX = 1, DM_EPI_DW = 2 };

struct DmaArgs {
  const unsigned short* A; const unsigned short* B;
  float* C; unsigned short* C16;
  const float* bias;           // FWD: per-column bias or null
  const float* mask;           // DX: C = mask[m][n] > 0 ? v : 0 (relu' of the layer below) or null
  const unsigned short* mask16;   // ... the bf16 twin of mask, read instead where there is one (half the bytes; same sign as the
                               //     fp32 value except 0 < x < 2^-134, which rounds to +0: stated in ff_hip.h)
  float* slots;                // DW: null = every k-slice adds its tile to C by atomics; else it stores it to slots[(tile * splitk + ks)][256][256] (dw_tile_slots, linear_gemm.h)
  const float* Af32;           // DW with db: the fp32 matrix behind A (same strides)
  float*       db;             // DW: db[m] += sum_k A(k, m) over this workgroup's share of its k-slice, from the fp32 values
                               //     [ref: src/ops/linear.cu:644-651], or null
  int64_t lda, ldb, ldc, ldmask;
  int M, N, K;
  int act;                     // FWD
  int add;                     // DX: C += v instead of C = v
  int splitk;                  // DW: k-slices per tile (grid = tiles x splitk)
  unsigned a_bytes, b_bytes;   // extents of the operands for the buffer descriptors
};

#define DM_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DM_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define DM_BARRIER() __builtin_amdgcn_s_barrier()

// inline asm on purpose: hipcc's waitcnt pass must not see the LDS-DMA, or it drains vmcnt(0) in front of every ds_read
__device__ __forceinline__ void dm_glds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

template <bool AKR, bool BKR, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_dma_kernel(const DmaArgs g) {
  extern __shared__ __attribute__((aligned(16))) char dm_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int c = lane & 15, q = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // ---- tile of this workgroup (n fastest; the workgroups of one XCD take neighbouring tiles) and its k range ----
  const unsigned nbx = (unsigned)((g.N + DM_BN - 1) / DM_BN), nby = (unsigned)((g.M + DM_BM - 1) / DM_BM), ntiles = nbx * nby;
  const unsigned total = gridDim.x, w = blockIdx.x;
  const unsigned xcd = w & 7u, loc = w >> 3, qq = total >> 3, rem = total & 7u;
  const unsigned nlin = xcd * qq + (xcd < rem ? xcd : rem) + loc;
  const unsigned tile = nlin % ntiles, ks = nlin / ntiles;          // ks: k-slice (splitk > 1)
  const unsigned by = tile / nbx, bx = tile - by * nbx;
  const int m0 = (int)by * DM_BM, n0 = (int)bx * DM_BN;
  const int nk_all = g.K / DM_BK;
  const int kt0 = (int)((int64_t)nk_all * ks / g.splitk), kt1 = (int)((int64_t)nk_all * (ks + 1) / g.splitk);
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.B), 0, g.b_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dm_lds;

  // ---- staging roles: the per-lane part of the source offset (bytes); piece and unit position go into the scalar offset ----
  unsigned voffA, voffB;
  {
    const int u = tid >> 3, j = (tid & 7) ^ ((u >> 1) & 7);                                     // k-contiguous: unit-row, source chunk
    const int kr = tid >> 4, jr = (tid & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));             // rows-are-k: k-row, source chunk
    voffA = AKR ? (unsigned)((kr * g.lda + jr * 8) * 2) : (unsigned)((u * g.lda + j * 8) * 2);
    voffB = BKR ? (unsigned)((kr * g.ldb + jr * 8) * 2) : (unsigned)((u * g.ldb + j * 8) * 2);
  }
  const unsigned istepA = (unsigned)((AKR ? 32 : 64) * g.lda * 2), istepB = (unsigned)((BKR ? 32 : 64) * g.ldb * 2);
  auto stage = [&](const int unit, const int kt) {          // kt relative to kt0; beyond the range: loads that return 0 without touching memory
    const bool isA = unit < 2;
    const int hi = unit & 1;
    const int64_t ld = isA ? g.lda : g.ldb;
    const int o0 = (isA ? m0 : n0) + hi * 128;
    const bool kr = isA ? AKR : BKR;
    unsigned soff;
    if (kt >= nk) soff = isA ? g.a_bytes : g.b_bytes;
    else soff = kr ? (unsigned)(((int64_t)(kt0 + kt) * DM_BK * ld + o0) * 2) : (unsigned)(((int64_t)o0 * ld + (int64_t)(kt0 + kt) * DM_BK) * 2);
    const unsigned dst = lds_base + (unsigned)((kt & 1) * DM_BUF + unit * DM_UNIT) + (unsigned)wave * 1024u;
    dm_glds16(isA ? voffA : voffB, isA ? rsA : rsB, dst, soff);
    dm_glds16(isA ? voffA : voffB, isA ? rsA : rsB, dst + 8192u, soff + (isA ? istepA : istepB));
  };

  // ---- fragment read offsets (bytes inside a unit) ----
  // k-contiguous: lane (c, q) reads chunk 4 s + q of unit-row base + 16 f + c; the swizzle term is (c >> 1) for every fragment
  const int kcA0 = (grp * 64 + c) * 128 + ((q ^ (c >> 1)) << 4), kcA1 = (grp * 64 + c) * 128 + (((4 + q) ^ (c >> 1)) << 4);
  const int kcB0 = (wc * 32 + c) * 128 + ((q ^ (c >> 1)) << 4), kcB1 = (wc * 32 + c) * 128 + (((4 + q) ^ (c >> 1)) << 4);
  // rows-are-k: lane (q; tq, tp) reads 8 bytes at k-row 32 s + 8 q + tq (+ 4), chunk (o >> 3) + (tp >> 1), o = first row / column of the fragment
  const int krX1 = (tq << 2) | (2 * (q & 1)), krX2 = krX1 | 1;
  const int krRow = (8 * q + tq) * 256 + 8 * (tp & 1);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  auto frag_kr = [&](const char* unit, int o, int s) -> bf16x8 {
    const int ch = (o >> 3) + (tp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(unit + s * 8192 + krRow + ((ch ^ krX1) << 4)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(unit + s * 8192 + krRow + 1024 + ((ch ^ krX2) << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto fragA = [&](const char* unit, int f, int s) -> bf16x8 {
    if (!AKR) return *reinterpret_cast<const bf16x8*>(unit + (s ? kcA1 : kcA0) + f * 2048);
    return frag_kr(unit, grp * 64 + f * 16, s);
  };
  auto fragB = [&](const char* unit, int f, int s) -> bf16x8 {
    if (!BKR) return *reinterpret_cast<const bf16x8*>(unit + (s ? kcB1 : kcB0) + f * 2048);
    return frag_kr(unit, wc * 32 + f * 16, s);
  };

  bf16x8 aLo[4][2], aHi[4][2], bLo[2][2], bHi[2][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mfma_block = [&](bf16x8 (&a)[4][2], bf16x8 (&b)[2][2], const int tm0, const int tn0) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int s = 0; s < 2; s++)      // operands swapped: a lane then holds 4 consecutive columns of one row
          acc[tm0 + tm][tn0 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn][s], a[tm][s], acc[tm0 + tm][tn0 + tn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  f32x4 bias_r[4];          // FWD: the bias of this lane's four column quads
#pragma unroll
  for (int tn = 0; tn < 4; tn++) {
    bias_r[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == DM_EPI_FWD) {
      const int cc = n0 + (tn >> 1) * 128 + wc * 32 + (tn & 1) * 16 + 4 * q;
      if (g.bias && cc < g.N) bias_r[tn] = *reinterpret_cast<const f32x4*>(g.bias + cc);
    }
  }

  // ---- prologue: k-tile 0 and the first two units of k-tile 1 ----
  stage(DM_BLO, 0); stage(DM_ALO, 0); stage(DM_BHI, 0); stage(DM_AHI, 0); stage(DM_BLO, 1); stage(DM_ALO, 1);
  if constexpr (EPI == DM_EPI_DW) {
    if (g.db) {
      // bias gradient = column sums of the fp32 dy (the bf16 twin would not do: the oracle sums fp32 values).  The k-tiles of this
      // k-slice are dealt round-robin to the tile columns, so every (k-tile, 256 rows of A^T) block is summed by exactly one
      // workgroup: 64 x 256 floats per block, 8 row groups x 64 column quads, then across the row groups through LDS.  While the
      // prologue's DMA pieces are in flight.
      const int cg = tid & 63, rg = tid >> 6;
      const int col = m0 + 4 * cg;
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col < g.M)
        for (int t = kt0 + (int)bx; t < kt1; t += (int)nbx) {
          const float* p = g.Af32 + ((int64_t)t * DM_BK + rg * 8) * g.lda + col;
#pragma unroll
          for (int r = 0; r < 8; r++) sum += *reinterpret_cast<const f32x4*>(p + (int64_t)r * g.lda);
        }
      f32x4* red = reinterpret_cast<f32x4*>(dm_lds + 2 * DM_BUF);
      red[rg * 64 + cg] = sum;
      __syncthreads();
      if (tid < 256) {
        const float* rf = reinterpret_cast<const float*>(red);
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) v += rf[r * 256 + tid];
        if (m0 + tid < g.M) atomicAdd(g.db + m0 + tid, v);
      }
      __syncthreads();
    }
  }
  DM_WAIT_VM(8);          // B-lo(0), A-lo(0) have landed
  DM_BARRIER();
  DM_FENCE();
  if (grp == 1) DM_BARRIER();      // group 1 runs half a phase behind group 0
  DM_FENCE();

  for (int t = 0; t < nk; t++) {
    const char* cb = dm_lds + (t & 1) * DM_BUF;
    // phase 1: A-lo x B-lo
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) bLo[f][s] = fragB(cb + DM_BLO * DM_UNIT, f, s);
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) aLo[f][s] = fragA(cb + DM_ALO * DM_UNIT, f, s);
    DM_FENCE();
    stage(DM_BHI, t + 1);
    DM_WAIT_VM(8);
    DM_BARRIER();
    DM_FENCE();
    mfma_block(aLo, bLo, 0, 0);
    DM_FENCE();
    DM_BARRIER();
    DM_FENCE();
    // phase 2: A-lo x B-hi
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) bHi[f][s] = fragB(cb + DM_BHI * DM_UNIT, f, s);
    DM_FENCE();
    stage(DM_AHI, t + 1);
    DM_WAIT_VM(8);
    DM_BARRIER();
    DM_FENCE();
    mfma_block(aLo, bHi, 0, 2);
    DM_FENCE();
    DM_BARRIER();
    DM_FENCE();
    // phase 3: A-hi x B-hi
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) aHi[f][s] = fragA(cb + DM_AHI * DM_UNIT, f, s);
    DM_FENCE();
    stage(DM_BLO, t + 2);
    DM_WAIT_VM(8);
    DM_BARRIER();
    DM_FENCE();
    mfma_block(aHi, bHi, 4, 2);
    DM_FENCE();
    DM_BARRIER();
    DM_FENCE();
    // phase 4: A-hi x B-lo
    stage(DM_ALO, t + 2);
    DM_WAIT_VM(8);
    DM_BARRIER();
    DM_FENCE();
    mfma_block(aHi, bLo, 4, 0);
    DM_FENCE();
    DM_BARRIER();
    DM_FENCE();
  }
  if (grp == 0) DM_BARRIER();
  DM_WAIT_VM(0);
  DM_BARRIER();           // every wave is past its last fragment read and its last DMA piece has landed: all of LDS is free

  // ---- epilogue: lane (c, q) holds C[row(tm) + c][col(tn) + 4 q + {0..3}].  Half of the wave's accumulators at a time (64 rows
  //      x 64 columns) goes through the wave's own 20 KB of LDS -- 16 writes, then rows back out: a store covers 8 rows x 128
  //      contiguous bytes (whole lines), an atomic instruction 2 rows x 128 (4-byte pieces 16 bytes apart run ~10x slower at
  //      the memory-side adders).  Rows padded to 272 B: conflict-free writes.
  constexpr int EP_LD = 272;
  char* blk = dm_lds + wave * 20480;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int rowb = m0 + half * 128 + grp * 64;          // tile row of the block's first row
    // the relu' mask of the rows this lane will store (dX; from the twin of x where it has one), fetched ahead of
    // the LDS round trip: rows p = 0..3 before the writes, p = 4..7 behind them
    f32x4 mkA[8], mkB[8];
    s16x4 mhA[8], mhB[8];
    auto mask_load = [&](f32x4 (&mk)[8], s16x4 (&mh)[8], const int p0) {
      if constexpr (EPI == DM_EPI_DX) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int p = p0 + (e >> 1), h = e & 1;
          const int row = rowb + p * 8 + (lane >> 3), col = n0 + h * 128 + wc * 32 + (lane & 7) * 4;
          const bool in = row < g.M && col < g.N;
          mk[e] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (g.mask16) {
            mh[e] = in ? *reinterpret_cast<const s16x4*>(g.mask16 + (int64_t)row * g.ldmask + col) : s16x4{0, 0, 0, 0};
          } else if (g.mask) {
            if (in) mk[e] = *reinterpret_cast<const f32x4*>(g.mask + (int64_t)row * g.ldmask + col);
          }
        }
      }
    };
    mask_load(mkA, mhA, 0);
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 4; tn++) {
        f32x4 v = acc[4 * half + tm][tn];
        if constexpr (EPI == DM_EPI_FWD) {
          v += bias_r[tn];
          v.x = act_apply(v.x, g.act); v.y = act_apply(v.y, g.act); v.z = act_apply(v.z, g.act); v.w = act_apply(v.w, g.act);
        }
        *reinterpret_cast<f32x4*>(blk + (tm * 16 + c) * EP_LD + ((tn >> 1) * 32 + (tn & 1) * 16 + 4 * q) * 4) = v;
      }
    // the wave's own block: no barrier, LDS operations of one wave execute in order
    if constexpr (EPI == DM_EPI_DW) {
      if (g.slots) {        // plain stores of whole 128-byte runs into the slice's own slot; the ordered pass follows the launch
        float* sl = g.slots + ((size_t)tile * (size_t)g.splitk + ks) * (size_t)(DM_BM * DM_BN) + (size_t)(half * 128 + grp * 64) * DM_BN + wc * 32;
        const int rr8 = lane >> 3, rc8 = lane & 7;
#pragma unroll
        for (int p = 0; p < 8; p++)
#pragma unroll
          for (int h = 0; h < 2; h++)
            *reinterpret_cast<f32x4*>(sl + (size_t)(p * 8 + rr8) * DM_BN + h * 128 + rc8 * 4) = *reinterpret_cast<const f32x4*>(blk + (p * 8 + rr8) * EP_LD + (h * 32 + rc8 * 4) * 4);
        continue;
      }
      const int rr = lane >> 5, rc = lane & 31;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int col = n0 + h * 128 + wc * 32 + rc;
#pragma unroll 8
        for (int p = 0; p < 32; p++) {
          const float v = *reinterpret_cast<const float*>(blk + (2 * p + rr) * EP_LD + (h * 32 + rc) * 4);
          const int row = rowb + 2 * p + rr;
          if (row < g.M && col < g.N) atomicAdd(g.C + (int64_t)row * g.ldc + col, v);
        }
      }
    } else {
      const int rr = lane >> 3, rc = lane & 7;
      mask_load(mkB, mhB, 4);
#pragma unroll
      for (int p = 0; p < 8; p++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          f32x4 v = *reinterpret_cast<const f32x4*>(blk + (p * 8 + rr) * EP_LD + (h * 32 + rc * 4) * 4);
          const int row = rowb + p * 8 + rr, col = n0 + h * 128 + wc * 32 + rc * 4;
          if (row < g.M && col < g.N) {
            float* cp = g.C + (int64_t)row * g.ldc + col;
            if constexpr (EPI == DM_EPI_DX) {
              const f32x4 mk = p < 4 ? mkA[(p & 3) * 2 + h] : mkB[(p & 3) * 2 + h];
              if (g.mask16) {        // a bf16 is > 0 iff its sign bit is clear and the rest is not zero: a signed 16-bit compare
                const s16x4 mh = p < 4 ? mhA[(p & 3) * 2 + h] : mhB[(p & 3) * 2 + h];
                v.x = mh[0] > 0 ? v.x : 0.f; v.y = mh[1] > 0 ? v.y : 0.f; v.z = mh[2] > 0 ? v.z : 0.f; v.w = mh[3] > 0 ? v.w : 0.f;
              } else if (g.mask) {
                v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
              }
              if (g.add) v += *reinterpret_cast<const f32x4*>(cp);
            }
            *reinterpret_cast<f32x4*>(cp) = v;
            if (g.C16) {
              const bf16x4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
              *reinterpret_cast<bf16x4*>(g.C16 + (int64_t)row * g.ldc + col) = t;
            }
          }
        }
    }
  }
}

}  // namespace

namespace ffh_gemm {

// 1: launched; 0: not this kernel's problem (nothing launched); < 0: error.  g.A16 / g.B16 (and g.C16) are set by the caller.
int launch_gemm_bf16_dma(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name) {
  static const int off = FFH_LAB_INT("FFH_BF16_NO_DMA", 0);     // A/B switch (tools/ab.sh)
  if (off || !g.A16 || !g.B16) return 0;
  if (form != BF16_FORM_FWD && form != BF16_FORM_DX && form != BF16_FORM_DW) return 0;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K % DM_BK || g.N % 8) return 0;
  if (g.colmap || g.act_y || g.fuse) return 0;
  const bool akr = form == BF16_FORM_DW, bkr = form != BF16_FORM_FWD;
  const int64_t lda = akr ? g.sAk : g.sAm, ldb = bkr ? g.sBk : g.sBn;
  if ((akr ? g.sAm : g.sAk) != 1 || (bkr ? g.sBn : g.sBk) != 1) return 0;
  if (lda % 8 || ldb % 8 || g.ldc % 4) return 0;
  if ((((uintptr_t)g.A16 | (uintptr_t)g.B16 | (uintptr_t)g.C) & 15) || ((uintptr_t)g.C16 & 7)) return 0;
  if (g.bias && ((uintptr_t)g.bias & 15)) return 0;
  if (g.mask && ((((uintptr_t)g.mask) & 15) || g.ldmask % 4)) return 0;
  if (form == BF16_FORM_DW && g.db && ((((uintptr_t)g.A) & 15) || g.M % 4)) return 0;
  const int64_t a_bytes = ((akr ? (int64_t)(g.K - 1) : (int64_t)(g.M - 1)) * lda + (akr ? g.M : g.K)) * 2;
  const int64_t b_bytes = ((bkr ? (int64_t)(g.K - 1) : (int64_t)(g.N - 1)) * ldb + (bkr ? g.N : g.K)) * 2;
  if (a_bytes >= (1LL << 31) || b_bytes >= (1LL << 31)) return 0;              // 32-bit buffer offsets, with room for the run-ahead
  const int64_t tiles = (int64_t)((g.M + DM_BM - 1) / DM_BM) * ((g.N + DM_BN - 1) / DM_BN);
  const int nk = g.K / DM_BK;
  int splitk = 1;
  if (form == BF16_FORM_DW) {
    if (g.epi != EPI_ATOMIC || c->deterministic) return 0;       // the k-slices of a tile meet by atomics
    // (round 5: from two tiles on -- 512 x 256 as 2 tiles x 64 slices takes 48.8 us where the 128 x 128 kernel took 93.8 and the fp32
    //  kernel takes 79.7: with eight as the limit this was the one layer of the step that tensor-op mode made slower)
    static const int min_tiles = FFH_LAB_INT("FFH_BF16_DMA_DW_MIN_TILES", 2);     // A/B switch
    if (tiles < min_tiles) return 0;                             // a single tile: at most 64 slices, a quarter of the chip
    // the split with the least estimated time: rounds of one workgroup per CU x k-tiles per slice (~1 us each at the main
    // loop's rate) + the slices' atomic traffic at the memory-side adders' ~1.3 TB/s (every slice adds a whole tile).  At batch
    // 32768: 3456 x 1024 -> 4 slices (224 workgroups, one round), 1024 x 1024 -> 16, 1024 x 512 -> 16
    static const int split_env = FFH_LAB_INT("FFH_BF16_DMA_SPLIT", 0);     // A/B switch
    const double tile_bytes = (double)DM_BM * DM_BN * 4;
    double best = 1e30;
    for (int sp = 1; sp <= 64 && sp * 4 <= nk; sp++) {
      const int64_t nb = tiles * sp;
      if (nb * 2 < c->num_cus) continue;
      const double rounds = (double)((nb + c->num_cus - 1) / c->num_cus);
      const double t = rounds * (double)((nk + sp - 1) / sp) * 1.0 + (double)nb * tile_bytes / 1.3e6;
      if (t < best) { best = t; splitk = sp; }
    }
    if (best == 1e30) return 0;
    if (split_env > 0 && split_env * 4 <= nk) splitk = split_env;
  } else {
    if (g.epi != EPI_STORE && g.epi != EPI_ADD) return 0;
    if (form == BF16_FORM_FWD && g.epi != EPI_STORE) return 0;
    // From half as many tiles as CUs on (round 5; before: as many).  Alone the 128 x 128 kernel is the faster one on such a shape (32768 x 512 ->
    // 256 forward 27 vs 34 us, 8192 x 1024 -> 1024 36 vs 43), in the step it is the slower: its workgroups put MFMA-dense waves on every CU and
    // the gather on the side stream stands still beside them, 128 tiles of this kernel leave it half the chip -- tensor-op
    // step 2.002-2.012 -> 1.973-1.984 ms at 32768 samples, 1.309 -> 1.219 at 16384, 0.959 -> 0.917 at 8192.
    static const int min_pct = FFH_LAB_INT("FFH_BF16_DMA_MIN_TILES_PCT", 50);      // A/B switch: least number of tiles, in per cent of the CUs
    if (tiles * 100 < (int64_t)c->num_cus * min_pct) return 0;   // fewer: linear_bf16.hip's 128 x 128 tiles
  }
  if (tiles * splitk >= (1LL << 31)) return 0;
  DmaArgs a{};
  a.A = g.A16; a.B = g.B16; a.C = g.C; a.C16 = form == BF16_FORM_DW ? nullptr : g.C16;
  a.bias = form == BF16_FORM_FWD ? g.bias : nullptr; a.mask = form == BF16_FORM_DX ? g.mask : nullptr;
  a.Af32 = g.A; a.db = form == BF16_FORM_DW ? g.db : nullptr;
  if (form == BF16_FORM_DW) a.slots = dw_tile_slots(c, s, tiles, splitk, g.ldc);
  static const int mask_fp32 = FFH_LAB_INT("FFH_BF16_MASK_FP32", 0);      // A/B switch
  a.mask16 = (a.mask && !mask_fp32) ? ffh_mirror_of(c, g.mask, (size_t)((int64_t)(g.M - 1) * g.ldmask + g.N) * 4) : nullptr;
  if (a.mask16 && ((uintptr_t)a.mask16 & 7)) a.mask16 = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = g.ldc; a.ldmask = g.ldmask;
  a.M = g.M; a.N = g.N; a.K = g.K; a.act = g.act; a.add = g.epi == EPI_ADD; a.splitk = splitk;
  a.a_bytes = (unsigned)a_bytes; a.b_bytes = (unsigned)b_bytes;
  const unsigned grid = (unsigned)(tiles * splitk);
#define FFH_DMA_LAUNCH(AKR, BKR, EPI)                                                                            \
  {                                                                                                              \
    auto kern = gemm_bf16_dma_kernel<AKR, BKR, EPI>;                                                             \
    static const bool ok = glds_set_lds(kern, DM_LDS);                                                           \
    if (!ok) return 0;                                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), DM_LDS, as_stream(s), a);                                    \
  }
  if (form == BF16_FORM_FWD) FFH_DMA_LAUNCH(false, false, DM_EPI_FWD)
  else if (form == BF16_FORM_DX) FFH_DMA_LAUNCH(false, true, DM_EPI_DX)
  else FFH_DMA_LAUNCH(true, true, DM_EPI_DW)
#undef FFH_DMA_LAUNCH
  if (a.slots) launch_dw_tile_reduce(a.slots, g.C, g.ldc, g.M, g.N, tiles, splitk, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  { char tok[96]; snprintf(tok, sizeof tok, "%s|bf16_dma_256x256_twins|splitk=%d%s", name, splitk, a.slots ? "|slots" : ""); ffh_route_add(c, tok); }
  return 1;
}

}  // namespace ffh_gemm
