// linear_x3_dma.hip -- the fp32-accurate split GEMM (FFH_MATH_FP32_SPLIT_BF16X3) fed from producer-kept three-plane images
// (ffh_ctx_bf16x3_mirror_set): the operands' bf16 terms come straight into LDS by LDS-DMA, no vector instruction touches an operand.
//
// Every fp32 element x is three bfloat16 terms x1 + x2 + x3 kept in the I32 image (ff_hip.h): 32 consecutive elements <-> 192 bytes
// [x1 of the 32 | x2 of the 32 | x3 of the 32].  A row of a matrix with leading dimension ld (a multiple of 32) is 6 * ld bytes; a k-tile of
// 32 of a k-contiguous row is 192 contiguous bytes, 128 columns of a k-row 768.
//
// 256 x 256 x 32 tiles, one workgroup of 8 waves per CU (two per SIMD), v_mfma_f32_16x16x32_bf16, six products per k-step
// (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1: small terms first), operands global -> LDS by buffer_load_dwordx4 ... lds into a ring of six
// 24 KB units.
//   * The waves are 2 (row groups) x 4 (column groups); a wave owns rows {64g..64g+63} + {128+64g..} and the 64 CONTIGUOUS columns
//     {64c..64c+63} of the tile (so that a row of its output is 256 / 384 contiguous, line-aligned bytes of C / C's image): 8 x 4 accumulators
//     of 16 x 16.  A k-tile is four PHASES of 48 MFMAs (one quadrant of the wave's output x 32 k x six products):
//        0: A-lo x B-lo   1: A-lo x B-hi   2: A-hi x B-hi   3: A-hi x B-lo        (B-lo / B-hi: the first / second 32 of every wave's 64 columns)
//     and every phase reads exactly ONE unit (128 rows / columns x 32 k x three planes) into registers:
//        0: A-lo(t)   1: B-hi(t)   2: A-hi(t)   3: B-lo(t + 1)  (into the registers B-hi(t) has just left)
//     so the units form one sequence s = 0, 1, 2, ...: B-lo(0), A-lo(0), B-hi(0), A-hi(0), B-lo(1), ...; unit s lives in ring slot s % 6, is
//     read in phase s - 1, was issued in phase s - 5 and waited for (vmcnt(9): three units stay in flight) in phase s - 3; a phase is
//         { the unit's fragment reads; 3 DMA pieces of unit s + 4; s_waitcnt vmcnt(9); s_barrier; 48 MFMAs at raised priority; s_barrier }
//   * The two row groups run HALF A PHASE APART (group 1 passes one extra barrier at the start), as in linear_bf16_dma.hip: while one wave
//     of a SIMD issues its MFMAs the other issues its reads and DMA pieces and waits at the barrier.  A slot is restaged two barrier intervals
//     after its last read at the earliest; a unit is read one barrier after the last wave's wait for it.
//   * LDS images are lane-linear (the DMA writes base + 16 * lane); the swizzles live on the SOURCE address and in the reads:
//       k-contiguous unit: 128 rows x 192 B [plane][4 chunks of 16 B], chunk j of row u at slot j ^ SW[(u >> 2) & 3], SW = {0, 2, 3, 1}: a
//         fragment (16 rows x 32 k of one plane) is one conflict-free ds_read_b128 per lane; a DMA piece covers 5 1/3 whole rows (192 B runs);
//       rows-are-k unit: 32 k-rows x 768 B [plane][16 chunks], chunk j (8 columns) of k-row r at slot j ^ (((r & 3) << 2) | ((r >> 2) & 3)): a
//         fragment is two ds_read_b64_tr_b16 (4 k x 16 columns each, transposed on the way out); a DMA piece covers 1 1/3 k-rows (768 B runs).
//   * Rows / columns beyond the matrix: the buffer descriptor starts at the tile and ends with the operand, so loads past it return 0
//     and touch nothing; columns past N of a rows-are-k operand read the next row's values into accumulators nobody stores.
//   * Epilogue through LDS as in linear_bf16_dma.hip (the wave's own 20 KB, half of its accumulators at a time): fp32 rows out in 256-byte
//     runs; then the image of the block in IMAGE order -- a row's 64 columns are 384 contiguous bytes = 24 chunks, lane L of pass `it`
//     computes the one plane of the 8 columns its chunk holds: every store instruction writes whole 128-byte lines.
// Measured (tools/lab/gemm_x3_lab.hip, steady clocks, 32768 samples): 3456 -> 1024 forward 828 us = 280 TFLOP/s fp32-equivalent (the
// split-in-kernel form: 1,098 us), weight gradient 903 us = 257 (1,241), 1024 -> 1024 forward 290 us = 237.  What bounds it: the matrix pipe --
// 48 MFMAs per phase back to back (SQ_VALU_MFMA_BUSY 86 % of the launch, the rest is the epilogue) at the ~1.9 GHz the chip holds under
// this load; non-MFMA vector instructions per MFMA: 0.11 (profiles/r06_pmc_x3_dma_lab.txt).
//
// Arithmetic: the same six exact products per 32-deep k-step, summed in fp32 in the same order as the split-in-kernel form
// (linear_bf16.hip) -- forward and data-gradient results are bit-identical with and without images.
// Replaces cublasSgemm [ref: src/ops/linear.cu:436-453,624-659] in this build's opt-in math mode; precedent for a faster math mode behind
// a handle switch: src/runtime/model.cu:81-83.
#include "linear_gemm.h"

#include <stdlib.h>

using namespace ffh_gemm;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int X_BM = 256, X_BN = 256, X_BK = 32;
constexpr int X_UNIT = 24576;                    // 128 rows (columns) x 32 k x 3 planes x 2 B
constexpr int X_NS = 6;                          // ring slots
constexpr int X_RED = X_NS * X_UNIT;             // 8 KB behind the ring: the bias-gradient reduction of the dW prologue
constexpr int X_LDS = 163840;                    // the epilogue uses all of it (20 KB per wave)

enum { X_EPI_FWD = 0, X_EPI_DX = 1, X_EPI_DW = 2 };

struct X3Args {
  const char* A; const char* B;      // I32 images of the operands' (0, 0) elements (each starts a group)
  float* C; char* C3;                // the fp32 output and its image (or null)
  const float* bias;                 // FWD: per-column bias or null
  const float* mask;                 // DX: C = mask[m][n] > 0 ? v : 0 (relu' of the layer below) or null
  const char* mask3;                 // ... the image of mask, whose first term is read instead where there is one (half the bytes; same sign as
                                     //     the fp32 value except 0 < x < 2^-134, which rounds to +0: stated in ff_hip.h)
  float* slots;                      // DW: null = every k-slice adds its tile to C by atomics; else slice ks of tile t stores it to slots[(t * splitk + ks)][256][256]
                                     //     (x3_dw_reduce_kernel then adds the slices in order: no atomics on C, the same bits every run)
  const float* Af32; float* db;      // DW: db[m] += sum_k A(k, m) over this workgroup's share of its k-slice, from the fp32 values
                                     //     [ref: src/ops/linear.cu:644-651], or null
  int64_t lda, ldb, ldc, ldmask;     // elements
  int M, N, K;
  int act, add, splitk;
  uint32_t a_bytes, b_bytes;         // extents of the operands' images, from A / B
};

#define X_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define X_BARRIER() __builtin_amdgcn_s_barrier()

// inline asm on purpose: hipcc's waitcnt pass must not see the LDS-DMA, or it drains vmcnt(0) in front of every ds_read
__device__ __forceinline__ void x_glds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

template <bool AKR, bool BKR, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_x3_dma_kernel(const X3Args g) {
  extern __shared__ __attribute__((aligned(16))) char x3_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int c = lane & 15, q = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // ---- tile of this workgroup (n fastest; the workgroups of one XCD take neighbouring tiles) and its k range ----
  const unsigned nbx = (unsigned)((g.N + X_BN - 1) / X_BN), nby = (unsigned)((g.M + X_BM - 1) / X_BM), ntiles = nbx * nby;
  const unsigned total = gridDim.x, w = blockIdx.x;
  const unsigned xcd = w & 7u, loc = w >> 3, qq = total >> 3, rem = total & 7u;
  const unsigned nlin = xcd * qq + (xcd < rem ? xcd : rem) + loc;
  const unsigned tile = nlin % ntiles, ks = nlin / ntiles;
  const unsigned by = tile / nbx, bx = tile - by * nbx;
  const int m0 = (int)by * X_BM, n0 = (int)bx * X_BN;
  const int nk_all = g.K / X_BK;
  const int kt0 = (int)((int64_t)nk_all * ks / g.splitk), kt1 = (int)((int64_t)nk_all * (ks + 1) / g.splitk);
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  // ---- buffer descriptors based at the tile's first row (k-contiguous) / first column (rows-are-k): an offset past the operand's end is
  //      out of range, returns 0 and touches nothing ----
  const int64_t lda6 = g.lda * 6, ldb6 = g.ldb * 6;
  const uint32_t a_org = __builtin_amdgcn_readfirstlane(AKR ? (uint32_t)(m0 / 32) * 192u : (uint32_t)((int64_t)m0 * lda6));
  const uint32_t b_org = __builtin_amdgcn_readfirstlane(BKR ? (uint32_t)(n0 / 32) * 192u : (uint32_t)((int64_t)n0 * ldb6));
  const uint32_t a_rec = __builtin_amdgcn_readfirstlane(a_org < g.a_bytes ? g.a_bytes - a_org : 0u), b_rec = __builtin_amdgcn_readfirstlane(b_org < g.b_bytes ? g.b_bytes - b_org : 0u);
  auto uniform_ptr = [](const char* p) {      // the descriptor words must be SGPRs for the inline-asm DMA: spell the uniformity out
    const uint64_t u = (uint64_t)p;
    return (char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u));
  };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(g.A + a_org), 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(g.B + b_org), 0, b_rec, 0x00020000);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)x3_lds;

  // ---- staging roles: piece i of this wave is LDS bytes [wave * 3072 + i * 1024, + 1024) of the unit, lane-linear ----
  unsigned voffA[3], voffB[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int idx = 64 * i + lane;                     // 16-byte chunk inside the wave's 3 KB
    {   // k-contiguous: 16 rows x [plane][4 chunks]
      const int r = 16 * wave + idx / 12, sl = idx % 12, pl = sl >> 2, cs = sl & 3;
      const int j = cs ^ ((0x1320 >> (4 * ((r >> 2) & 3))) & 3);
      if (!AKR) voffA[i] = (unsigned)(r * lda6 + pl * 64 + j * 16);
      if (!BKR) voffB[i] = (unsigned)((64 * (r >> 5) + (r & 31)) * ldb6 + pl * 64 + j * 16);
    }
    {   // rows-are-k: 4 k-rows x [plane][16 chunks]
      const int kr = 4 * wave + idx / 48, sl = idx % 48, pl = sl >> 4, cs = sl & 15;
      const int j = cs ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
      if (AKR) voffA[i] = (unsigned)(kr * lda6 + (j >> 2) * 192 + pl * 64 + (j & 3) * 16);
      if (BKR) voffB[i] = (unsigned)(kr * ldb6 + (j >> 2) * 384 + pl * 64 + (j & 3) * 16);
    }
  }
  // one unit: operand (A / B), half (rows / columns 0..127 or 128..255 of the tile), k-tile kt relative to kt0, into ring slot `slot`
  auto stage = [&](const bool isA, const int hi, const int kt, const int slot) {
    const bool kr = isA ? AKR : BKR;
    const int64_t ld6 = isA ? lda6 : ldb6;
    unsigned soff;
    if (kt >= nk) soff = isA ? a_rec : b_rec;
    else soff = kr ? (unsigned)((int64_t)(kt0 + kt) * X_BK * ld6 + hi * (isA ? 768 : 192)) : (unsigned)((int64_t)hi * (isA ? 128 : 32) * ld6 + (int64_t)(kt0 + kt) * 192);
    const unsigned dst = lds_base + (unsigned)(slot * X_UNIT) + (unsigned)wave * 3072u;
#pragma unroll
    for (int i = 0; i < 3; i++) x_glds16(isA ? voffA[i] : voffB[i], isA ? rsA : rsB, dst + 1024u * i, soff);
  };

  // ---- fragment read offsets (bytes inside a unit) ----
  const int swc = (0x1320 >> (4 * ((c >> 2) & 3))) & 3;
  const int kcA = (grp * 64 + c) * 192 + ((q ^ swc) << 4), kcB = (wc * 32 + c) * 192 + ((q ^ swc) << 4);
  // rows-are-k: lane (q; tq, tp) fetches 8 bytes of k-row 8 q + tq (and of the row four below) at chunk ch = 2 F + (tp >> 1), F = the fragment's
  // number among the unit's eight 16-column groups; with the swizzle the chunk's slot is 2 (F ^ Y) + (tp >> 1), Y = (tq << 1) | (q & 1) -- and one
  // bit less of (tp >> 1) for the second row: the eight (four) fragment addresses of a unit are base + 32 ((f ^ y) & 3 [& 1]), second row + a
  // per-lane constant.  Spelled out so that three registers per operand carry what would be twelve loop-invariant addresses.
  const int krY = (tq << 1) | (q & 1), krT1 = tp >> 1;
  const int krRow = (8 * q + tq) * 768 + 8 * (tp & 1) + 16 * krT1;
  const int krBaseA = krRow + 128 * (grp ^ (krY >> 2)), krBaseB = krRow + 64 * (wc ^ (krY >> 1));
  const int krDelta = 3072 + (krT1 ? -16 : 16);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  auto frag_kr = [&](const char* at, int p) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(at + p * 256));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(at + krDelta + p * 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto readA = [&](bf16x8 (&a)[4][3], const int slot) {
    const char* u = x3_lds + slot * X_UNIT;
    int y = krY & 3;
    if (AKR) asm volatile("" : "+v"(y));        // keeps the four XORs inside the loop (see above)
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        if (!AKR) a[f][p] = *reinterpret_cast<const bf16x8*>(u + kcA + f * 3072 + p * 64);
        else a[f][p] = frag_kr(u + krBaseA + ((f ^ y) << 5), p);
      }
  };
  auto readB = [&](bf16x8 (&b)[2][3], const int slot) {
    const char* u = x3_lds + slot * X_UNIT;
    int y = krY & 1;
    if (BKR) asm volatile("" : "+v"(y));
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        if (!BKR) b[f][p] = *reinterpret_cast<const bf16x8*>(u + kcB + f * 3072 + p * 64);
        else b[f][p] = frag_kr(u + krBaseB + ((f ^ y) << 5), p);
      }
  };

  bf16x8 a[4][3], bP[2][3], bQ[2][3];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the six products with i + j <= 4, small terms first; consecutive MFMAs go to different accumulators
  auto mfma_phase = [&](bf16x8 (&av)[4][3], bf16x8 (&bv)[2][3], const int tm0, const int tn0) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int pr = 0; pr < 6; pr++) {
      const int pa = pr == 0 ? 2 : (pr == 1 || pr >= 4 ? 0 : 1), pb = pr == 1 ? 2 : (pr == 2 || pr == 4 ? 1 : 0);
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)      // operands swapped: a lane then holds 4 consecutive columns of one row
          acc[tm0 + tm][tn0 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[tn][pb], av[tm][pa], acc[tm0 + tm][tn0 + tn], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: units 0..4 = B-lo(0), A-lo(0), B-hi(0), A-hi(0), B-lo(1) ----
  stage(false, 0, 0, 0); stage(true, 0, 0, 1); stage(false, 1, 0, 2); stage(true, 1, 0, 3); stage(false, 0, 1, 4);
  if constexpr (EPI == X_EPI_DW) {
    if (g.db) {
      const int cg = tid & 63, rg = tid >> 6;
      const int col = m0 + 4 * cg;
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col < g.M)
        for (int t = kt0 + (int)bx; t < kt1; t += (int)nbx) {
          const float* p = g.Af32 + ((int64_t)t * X_BK + rg * 4) * g.lda + col;
#pragma unroll
          for (int r = 0; r < 4; r++) sum += *reinterpret_cast<const f32x4*>(p + (int64_t)r * g.lda);
        }
      f32x4* red = reinterpret_cast<f32x4*>(x3_lds + X_RED);
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
  X_WAIT_VM(9);           // units 0 and 1 have landed
  X_BARRIER();
  X_FENCE();
  if (grp == 1) X_BARRIER();      // group 1 runs half a phase behind group 0
  X_FENCE();
  readB(bQ, 0);
  X_FENCE();

  int rs = 1, ws = 5;             // ring slots of the unit this phase reads / stages
  auto next = [](int s) { return s == X_NS - 1 ? 0 : s + 1; };
  // phases of k-tile t with B-lo(t) in bl; bo receives B-hi(t), then B-lo(t + 1)
  auto tile_body = [&](const int t, bf16x8 (&bl)[2][3], bf16x8 (&bo)[2][3]) {
    // phase 0: A-lo x B-lo
    readA(a, rs); rs = next(rs);
    X_FENCE();
    stage(true, 0, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bl, 0, 0);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 1: A-lo x B-hi
    readB(bo, rs); rs = next(rs);
    X_FENCE();
    stage(false, 1, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bo, 0, 2);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 2: A-hi x B-hi
    readA(a, rs); rs = next(rs);
    X_FENCE();
    stage(true, 1, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bo, 4, 2);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 3: A-hi x B-lo; B-lo(t + 1) into the registers B-hi(t) has left
    readB(bo, rs); rs = next(rs);
    X_FENCE();
    stage(false, 0, t + 2, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bl, 4, 0);
    X_FENCE(); X_BARRIER(); X_FENCE();
  };
  for (int t = 0; t < nk; t += 2) {
    tile_body(t, bQ, bP);
    if (t + 1 < nk) tile_body(t + 1, bP, bQ);
  }
  if (grp == 0) X_BARRIER();
  X_WAIT_VM(0);
  X_BARRIER();           // every wave is past its last fragment read and its last DMA piece has landed: all of LDS is free

  // ---- epilogue: as linear_bf16_dma.hip (half of the wave's accumulators at a time through the wave's own 20 KB of LDS) ----
  constexpr int EP_LD = 272;
  char* blk = x3_lds + wave * 20480;
  const int colw = n0 + 64 * wc;                 // the wave's 64 contiguous columns
  f32x4 bias_r[4];
#pragma unroll
  for (int tn = 0; tn < 4; tn++) {
    bias_r[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == X_EPI_FWD) {
      const int cc = colw + tn * 16 + 4 * q;
      if (g.bias && cc < g.N) bias_r[tn] = *reinterpret_cast<const f32x4*>(g.bias + cc);
    }
  }
  const int64_t ldc6 = g.ldc * 6, ldm6 = g.ldmask * 6;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int rowb = m0 + half * 128 + grp * 64;
    // the relu' mask of the rows this lane will store (dX; from plane 1 of x where it has planes), fetched ahead of the LDS round trip
    f32x4 mkA[8], mkB[8];
    s16x4 mhA[8], mhB[8];
    auto mask_load = [&](f32x4 (&mk)[8], s16x4 (&mh)[8], const int p0) {
      if constexpr (EPI == X_EPI_DX) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int row = rowb + (p0 + e) * 4 + (lane >> 4), col = colw + (lane & 15) * 4;
          const bool in = row < g.M && col < g.N;
          mk[e] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (g.mask3) {
            mh[e] = in ? *reinterpret_cast<const s16x4*>(g.mask3 + (int64_t)row * ldm6 + ffh_i32_off(col)) : s16x4{0, 0, 0, 0};
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
        if constexpr (EPI == X_EPI_FWD) {
          v += bias_r[tn];
          v.x = act_apply(v.x, g.act); v.y = act_apply(v.y, g.act); v.z = act_apply(v.z, g.act); v.w = act_apply(v.w, g.act);
        }
        *reinterpret_cast<f32x4*>(blk + (tm * 16 + c) * EP_LD + (tn * 16 + 4 * q) * 4) = v;
      }
    // the wave's own block: no barrier, LDS operations of one wave execute in order
    if constexpr (EPI == X_EPI_DW) {
      if (g.slots) {        // whole 256-byte runs of the slice's own slot: plain stores (ctx scratch, reserved per stream)
        float* sl = g.slots + ((size_t)tile * (size_t)g.splitk + ks) * (size_t)(X_BM * X_BN) + (size_t)(half * 128 + grp * 64) * X_BN + 64 * wc;
        const int rr4 = lane >> 4, rc4 = lane & 15;
#pragma unroll
        for (int p = 0; p < 16; p++)
          *reinterpret_cast<f32x4*>(sl + (size_t)(p * 4 + rr4) * X_BN + rc4 * 4) = *reinterpret_cast<const f32x4*>(blk + (p * 4 + rr4) * EP_LD + rc4 * 16);
        continue;
      }
      const int rr = lane >> 5, rc = lane & 31;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int col = colw + h * 32 + rc;
#pragma unroll 8
        for (int p = 0; p < 32; p++) {
          const float v = *reinterpret_cast<const float*>(blk + (2 * p + rr) * EP_LD + (h * 32 + rc) * 4);
          const int row = rowb + 2 * p + rr;
          if (row < g.M && col < g.N) atomicAdd(g.C + (int64_t)row * g.ldc + col, v);
        }
      }
    } else {
      const int rr = lane >> 4, rc = lane & 15;       // a store covers 4 rows x 256 contiguous bytes
      mask_load(mkB, mhB, 8);
#pragma unroll
      for (int p = 0; p < 16; p++) {
        f32x4 v = *reinterpret_cast<const f32x4*>(blk + (p * 4 + rr) * EP_LD + rc * 16);
        const int row = rowb + p * 4 + rr, col = colw + rc * 4;
        if (row < g.M && col < g.N) {
          float* cp = g.C + (int64_t)row * g.ldc + col;
          if constexpr (EPI == X_EPI_DX) {
            const f32x4 mk = p < 8 ? mkA[p & 7] : mkB[p & 7];
            if (g.mask3) {
              const s16x4 mh = p < 8 ? mhA[p & 7] : mhB[p & 7];
              v.x = mh[0] > 0 ? v.x : 0.f; v.y = mh[1] > 0 ? v.y : 0.f; v.z = mh[2] > 0 ? v.z : 0.f; v.w = mh[3] > 0 ? v.w : 0.f;
            } else if (g.mask) {
              v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
            if (g.add) v += *reinterpret_cast<const f32x4*>(cp);
          }
          *reinterpret_cast<f32x4*>(cp) = v;
          if constexpr (EPI == X_EPI_DX) { if (g.C3) *reinterpret_cast<f32x4*>(blk + (p * 4 + rr) * EP_LD + rc * 16) = v; }   // as stored, for the plane pass
        }
      }
      // the planes of the block, in the order of the plane image: a row's 64 columns are two groups = 384 contiguous bytes = 24 chunks of
      // 16 bytes (8 columns of one plane); lane L of pass `it` takes chunk it * 64 + L of the block's 64 x 24, computes that plane of its 8
      // columns and stores it -- every store instruction writes whole 128-byte lines (a half-written line costs a fill from memory: the
      // same bytes as 64-byte pieces per row took 120 us more per launch at 32768 x 1024)
      if (g.C3) {
        char* c3w = g.C3 + (int64_t)(colw >> 5) * 192;
#pragma unroll 4
        for (int it = 0; it < 24; it++) {
          const int L = it * 64 + lane, r = L / 24, ch = L - r * 24;
          const int grp32 = ch >= 12 ? 1 : 0, pl = (ch - 12 * grp32) >> 2, c8 = grp32 * 32 + (ch & 3) * 8;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(blk + r * EP_LD + c8 * 4);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(blk + r * EP_LD + c8 * 4 + 16);
          uint2 a1, a2, a3, b1, b2, b3;
          ffh_split_bf16x3(make_float4(v0.x, v0.y, v0.z, v0.w), a1, a2, a3); ffh_split_bf16x3(make_float4(v1.x, v1.y, v1.z, v1.w), b1, b2, b3);
          const uint4 o = pl == 0 ? make_uint4(a1.x, a1.y, b1.x, b1.y) : (pl == 1 ? make_uint4(a2.x, a2.y, b2.x, b2.y) : make_uint4(a3.x, a3.y, b3.x, b3.y));
          const int row = rowb + r;
          char* dst = c3w + (int64_t)row * ldc6 + ch * 16;
          if (row < g.M && colw + c8 < g.N) *reinterpret_cast<uint4*>(dst) = o;
        }
      }
    }
  }
}

// C[tile] += slot(tile, 0) + slot(tile, 1) + ... in slice order: one thread per four columns of one row of a tile
__global__ __launch_bounds__(256) void x3_dw_reduce_kernel(const float* __restrict__ slots, float* __restrict__ C, int64_t ldc, int M, int N, int nbx, int splitk) {
  const unsigned tile = blockIdx.x >> 6, part = blockIdx.x & 63u;                  // 64 workgroups of 256 threads x 4 floats per tile
  const unsigned e4 = part * 256u + threadIdx.x, r = e4 >> 6, c4 = (e4 & 63u) * 4u;
  const int by = (int)(tile / (unsigned)nbx), bx = (int)(tile - (unsigned)by * (unsigned)nbx);
  const int row = by * X_BM + (int)r, col = bx * X_BN + (int)c4;
  if (row >= M || col >= N) return;
  const float* sp = slots + (size_t)tile * (size_t)splitk * (size_t)(X_BM * X_BN) + (size_t)r * X_BN + c4;
  f32x4 s = *reinterpret_cast<const f32x4*>(sp);
  for (int k = 1; k < splitk; k++) s += *reinterpret_cast<const f32x4*>(sp + (size_t)k * (size_t)(X_BM * X_BN));
  float* cp = C + (int64_t)row * ldc + col;
  if (col + 3 < N) { *reinterpret_cast<f32x4*>(cp) = *reinterpret_cast<const f32x4*>(cp) + s; }
  else { for (int e = 0; e < 4 && col + e < N; e++) cp[e] += s[e]; }
}

}  // namespace

namespace ffh_gemm {

// the stream's slots for the k-slices of a weight gradient on 256 x 256 tiles (ffh_ctx_reserve_scratch), or null: no scratch on this stream / too many slices /
// more than one round of workgroups (with two rounds the first one's atomics run under the second one's MFMAs: 3456 x 1024 at 32768 samples 893 vs 906 us)
float* dw_tile_slots(const ffh_ctx* c, ffh_stream s, int64_t tiles, int splitk, int64_t ldc) {
  static const int no_slots = FFH_LAB_INT("FFH_X3_DW_NO_SLOTS", 0);      // A/B switches
  static const int slots_rounds = FFH_LAB_INT("FFH_X3_DW_SLOTS_MAX_ROUNDS", 1);
  if (no_slots || splitk <= 1 || tiles * splitk > kX3DwSlots || tiles * splitk > (int64_t)slots_rounds * c->num_cus || ldc % 4) return nullptr;
  for (int i = 0; i < c->nscratch; i++)
    if (c->scratch[i].stream == (void*)as_stream(s) && c->scratch[i].x3_slots) return c->scratch[i].x3_slots;
  return nullptr;
}
void launch_dw_tile_reduce(const float* slots, float* C, int64_t ldc, int M, int N, int64_t tiles, int splitk, ffh_stream s) {
  hipLaunchKernelGGL(x3_dw_reduce_kernel, dim3((unsigned)tiles * 64u), dim3(256), 0, as_stream(s), slots, C, ldc, M, N, (int)((N + X_BN - 1) / X_BN), splitk);
}

// 1: launched; 0: not this kernel's problem (nothing launched); < 0: error.  g.C3 (the image of C, or null) is set by the caller.
int launch_gemm_x3_dma(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name) {
  static const int off = FFH_LAB_INT("FFH_X3_NO_DMA", 0);     // A/B switch (tools/ab.sh)
  if (off || !ffh_split_mode(c) || g.a_not_twinned) return 0;
  if (form != BF16_FORM_FWD && form != BF16_FORM_DX && form != BF16_FORM_DW) return 0;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K % X_BK || g.N % 8) return 0;
  if (g.colmap || g.act_y || g.fuse) return 0;
  const bool akr = form == BF16_FORM_DW, bkr = form != BF16_FORM_FWD;
  const int64_t lda = akr ? g.sAk : g.sAm, ldb = bkr ? g.sBk : g.sBn;
  if ((akr ? g.sAm : g.sAk) != 1 || (bkr ? g.sBn : g.sBk) != 1) return 0;
  if (lda % 32 || ldb % 32 || g.ldc % 4) return 0;
  if (((uintptr_t)g.C & 15) || (g.bias && ((uintptr_t)g.bias & 15))) return 0;
  if (g.mask && ((((uintptr_t)g.mask) & 15) || g.ldmask % 4)) return 0;
  if (form == BF16_FORM_DW && g.db && ((((uintptr_t)g.A) & 15) || g.M % 4)) return 0;
  // both operands from their images: registered regions, each operand starting a 32-element group
  const int64_t a_rows = akr ? g.K : g.M, a_cols = akr ? g.M : g.K, b_rows = bkr ? g.K : g.N, b_cols = bkr ? g.N : g.K;
  const char* a3 = ffh_planes_of(c, g.A, (size_t)((a_rows - 1) * lda + a_cols) * 4);
  const char* b3 = ffh_planes_of(c, g.B, (size_t)((b_rows - 1) * ldb + b_cols) * 4);
  if (!a3 || !b3) return 0;
  const int64_t a_bytes = (a_rows - 1) * lda * 6 + (a_cols + 31) / 32 * 192, b_bytes = (b_rows - 1) * ldb * 6 + (b_cols + 31) / 32 * 192;
  if (a_bytes >= (1LL << 32) - (1 << 24) || b_bytes >= (1LL << 32) - (1 << 24)) return 0;       // 32-bit buffer offsets, with room for the run-ahead
  const int64_t tiles = (int64_t)((g.M + X_BM - 1) / X_BM) * ((g.N + X_BN - 1) / X_BN);
  const int nk = g.K / X_BK;
  int splitk = 1;
  if (form == BF16_FORM_DW) {
    if (g.epi != EPI_ATOMIC || c->deterministic) return 0;       // the k-slices of a tile meet by atomics
    static const int min_tiles = FFH_LAB_INT("FFH_X3_DMA_DW_MIN_TILES", 2);     // A/B switch
    if (tiles < min_tiles) return 0;                             // a single tile: at most 64 slices, a quarter of the chip
    // the split with the least estimated time: rounds of one workgroup per CU x k-tiles per slice (~3 us each: 192 MFMAs per wave) + the
    // slices' atomic traffic at the memory-side adders' ~1.3 TB/s (every slice adds a whole tile)
    static const int split_env = FFH_LAB_INT("FFH_X3_DMA_SPLIT", 0);     // A/B switch
    const double tile_bytes = (double)X_BM * X_BN * 4;
    double best = 1e30;
    for (int sp = 1; sp <= 64 && sp * 8 <= nk; sp++) {
      const int64_t nb = tiles * sp;
      if (nb * 2 < c->num_cus) continue;
      const double rounds = (double)((nb + c->num_cus - 1) / c->num_cus);
      const double t = rounds * (double)((nk + sp - 1) / sp) * 3.0 + (double)nb * tile_bytes / 1.3e6;
      if (t < best) { best = t; splitk = sp; }
    }
    if (best == 1e30) return 0;
    if (split_env > 0 && split_env * 4 <= nk) splitk = split_env;
  } else {
    if (g.epi != EPI_STORE && g.epi != EPI_ADD) return 0;
    if (form == BF16_FORM_FWD && g.epi != EPI_STORE) return 0;
    // at least three tiles per four CUs: with 128 tiles (8192 x 3456 -> 1024 forward) this kernel takes 307-322 us where the split-in-kernel
    // 128 x 128 form, whose 512 tiles fill the chip, takes 280; 8192 x 1024 -> 1024: 108-121 against 93; with 224 tiles (4096 x 3456 -> 1024 data
    // gradient) it takes 123-145 us against 196-219 (profiles/r06_microbench_x3_images.txt): a tile runs at 1/256 of this kernel's ~290 TF, the
    // 128 x 128 form reaches 150-210 on the whole chip
    static const int min_pct = FFH_LAB_INT("FFH_X3_DMA_MIN_TILES_PCT", 75);       // A/B switch: least number of tiles, in per cent of the CUs
    if (tiles * 100 < (int64_t)c->num_cus * min_pct) return 0;
  }
  if (tiles * splitk >= (1LL << 31)) return 0;
  X3Args a{};
  a.A = a3; a.B = b3; a.C = g.C;
  a.C3 = (form != BF16_FORM_DW && g.ldc % 32 == 0) ? ffh_planes_of(c, g.C, (size_t)((int64_t)(g.M - 1) * g.ldc + g.N) * 4) : nullptr;
  a.bias = form == BF16_FORM_FWD ? g.bias : nullptr; a.mask = form == BF16_FORM_DX ? g.mask : nullptr;
  a.mask3 = (a.mask && g.ldmask % 32 == 0) ? ffh_planes_of(c, g.mask, (size_t)((int64_t)(g.M - 1) * g.ldmask + g.N) * 4) : nullptr;
  a.Af32 = g.A; a.db = form == BF16_FORM_DW ? g.db : nullptr;
  // the weight gradient's k-slices through the stream's slots instead of atomics: 129 MB of float atomics (3456 x 1024, 9 slices) take ~100 us
  // at the memory-side adders, plain stores + an ordered pass over them a third of that -- and the result no longer depends on arrival order
  if (form == BF16_FORM_DW) a.slots = dw_tile_slots(c, s, tiles, splitk, g.ldc);
  a.lda = lda; a.ldb = ldb; a.ldc = g.ldc; a.ldmask = g.ldmask;
  a.M = g.M; a.N = g.N; a.K = g.K; a.act = g.act; a.add = g.epi == EPI_ADD; a.splitk = splitk;
  a.a_bytes = (uint32_t)a_bytes; a.b_bytes = (uint32_t)b_bytes;
  const unsigned grid = (unsigned)(tiles * splitk);
#define FFH_X3_LAUNCH(AKR, BKR, EPI)                                                                              \
  {                                                                                                              \
    auto kern = gemm_x3_dma_kernel<AKR, BKR, EPI>;                                                               \
    static const bool ok = glds_set_lds(kern, X_LDS);                                                            \
    if (!ok) return 0;                                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), X_LDS, as_stream(s), a);                                     \
  }
  if (form == BF16_FORM_FWD) FFH_X3_LAUNCH(false, false, X_EPI_FWD)
  else if (form == BF16_FORM_DX) FFH_X3_LAUNCH(false, true, X_EPI_DX)
  else FFH_X3_LAUNCH(true, true, X_EPI_DW)
#undef FFH_X3_LAUNCH
  if (a.slots) launch_dw_tile_reduce(a.slots, g.C, g.ldc, g.M, g.N, tiles, splitk, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  { char tok[112]; snprintf(tok, sizeof tok, "%s|x3_dma_256x256_planes%s|splitk=%d%s", name, a.C3 ? "+image" : "", splitk, a.slots ? "|slots" : ""); ffh_route_add(c, tok); }
  // the image of C where it has one the kernel could not write in place (C not at a group start, ldc not a multiple of 32)
  if (form != BF16_FORM_DW && !a.C3) {
    int col0 = 0;
    if (ffh_planes_of(c, g.C, (size_t)((int64_t)(g.M - 1) * g.ldc + g.N) * 4, &col0)) {
      const int rc = ffh_convert_f32_to_bf16x3(c, g.C, g.M, g.N, g.ldc, s);
      if (rc) return rc;
    }
  }
  return 1;
}

}  // namespace ffh_gemm
