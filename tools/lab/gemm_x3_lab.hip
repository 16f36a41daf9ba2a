// gemm_x3_lab.hip -- stand-alone lab for the fp32-accurate split GEMM fed from producer-kept bf16 planes (not part of the product;
// the kernel moves to csrc/linear_x3_dma.hip once it measures).
//
// Operands: every fp32 element x is kept as three bfloat16 terms x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)) in
// the I32 plane image: 32 consecutive fp32 elements <-> 192 bytes [x1 of the 32 | x2 of the 32 | x3 of the 32].  A row of a matrix with
// leading dimension ld (a multiple of 32) is 6 * ld bytes; a k-tile of 32 of a k-contiguous row is 192 contiguous bytes, 128 columns of a
// k-row are 768 contiguous bytes.
//
// Kernel: 256 x 256 x 32 tiles, one workgroup of 8 waves per CU (two per SIMD), v_mfma_f32_16x16x32_bf16, six products per k-step
// (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1: small terms first), operands global -> LDS by LDS-DMA into a ring of six 24 KB units.
//   * The waves are 2 (row groups) x 4 (column groups); a wave owns rows {64g..} + {128+64g..} and columns {32c..} + {128+32c..} of the
//     tile: 8 x 4 accumulators of 16 x 16.  A k-tile is four PHASES of 48 MFMAs (one quadrant x 32 k x six products):
//        0: A-lo x B-lo   1: A-lo x B-hi   2: A-hi x B-hi   3: A-hi x B-lo
//     and every phase reads exactly ONE unit (128 rows / columns x 32 k x three planes) into registers:
//        0: A-lo(t)   1: B-hi(t)   2: A-hi(t)   3: B-lo(t+1)  (into the register set B-hi(t) has just left)
//     so the units form one sequence s = 0, 1, 2, ...: B-lo(0), A-lo(0), B-hi(0), A-hi(0), B-lo(1), ...; unit s lives in ring slot s % 6,
//     is read in phase s - 1 and was issued in phase s - 5; a phase is
//         { the unit's fragment reads; 3 DMA pieces of unit s + 4; s_waitcnt vmcnt(9); s_barrier; 48 MFMAs; s_barrier }
//   * The two row groups run half a phase apart (as in linear_bf16_dma.hip).
//   * LDS images are lane-linear; swizzles live on the source address and in the reads:
//       k-contiguous unit: 128 rows x 192 B [plane][4 chunks of 16 B], chunk j of row u at slot j ^ SW[(u >> 2) & 3], SW = {0, 2, 3, 1}
//       rows-are-k unit:  32 k-rows x 768 B [plane][16 chunks], chunk j of k-row r at slot j ^ (((r & 3) << 2) | ((r >> 2) & 3))
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_x3_lab.hip -o tools/lab/gemm_x3_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>
#include <time.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------------------------
// the split and the I32 plane image
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4 v) {
  const bf16x2 lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}
__device__ __forceinline__ float bf16_lo_as_f32(const unsigned p) { unsigned r; asm("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float bf16_hi_as_f32(const unsigned p) { unsigned r; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(p)); return __uint_as_float(r); }
__device__ __forceinline__ float sub_f32(const float a, const float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 residual_f32x4(const f32x4 v, const uint2 p) {
  return f32x4{sub_f32(v.x, bf16_lo_as_f32(p.x)), sub_f32(v.y, bf16_hi_as_f32(p.x)), sub_f32(v.z, bf16_lo_as_f32(p.y)), sub_f32(v.w, bf16_hi_as_f32(p.y))};
}
__device__ __forceinline__ void split_bf16x3(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
  p1 = pack_bf16x4(v);
  const f32x4 r = residual_f32x4(v, p1);
  p2 = pack_bf16x4(r);
  p3 = pack_bf16x4(residual_f32x4(r, p2));
}
// byte offset of plane 0 of element e (a multiple of 4) in the I32 image of a buffer whose element 0 starts a group
__device__ __host__ __forceinline__ int64_t i32_off(int64_t e) { return (e >> 5) * 192 + (e & 31) * 2; }

// rows x cols fp32 (leading dimension ld, a multiple of 32; cols a multiple of 4) -> I32 planes
__global__ __launch_bounds__(256) void split3_kernel(char* __restrict__ dst, const float* __restrict__ src, int64_t rows, int cols, int64_t ld) {
  const int64_t per_row = cols / 4, total = rows * per_row, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / per_row; const int c = (int)(i - r * per_row) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld + c);
    uint2 p1, p2, p3; split_bf16x3(v, p1, p2, p3);
    char* d = dst + r * ld * 6 + i32_off(c);
    *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + 64) = p2; *reinterpret_cast<uint2*>(d + 128) = p3;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int X_BM = 256, X_BN = 256, X_BK = 32;
constexpr int X_UNIT = 24576;                    // 128 rows (columns) x 32 k x 3 planes x 2 B
constexpr int X_NS = 6;                          // ring slots
constexpr int X_RED = X_NS * X_UNIT;             // 8 KB behind the ring: the bias-gradient reduction of the dW prologue
constexpr int X_LDS = 163840;                    // the epilogue uses all of it (20 KB per wave)

enum { X_EPI_FWD = 0, X_EPI_DX = 1, X_EPI_DW = 2 };
enum { X_ACT_NONE = 0, X_ACT_RELU = 1 };

struct X3Args {
  const char* A; const char* B;      // I32 planes of the operands' (0, 0) elements
  float* C; char* C3;                // the fp32 output and its planes (or null)
  const float* bias;                 // FWD
  const float* mask;                 // DX: C = mask[m][n] > 0 ? v : 0, or null
  const char* mask3;                 // ... plane 1 of mask's I32 image, read instead where there is one
  const float* Af32; float* db;      // DW: db[m] += sum_k A(k, m) from the fp32 values, or null
  int64_t lda, ldb, ldc, ldmask;     // elements
  int M, N, K;
  int act, add, splitk;
  int diag;                          // lab: 1 no DMA in the loop, 2 no fragment reads, 3 no MFMAs, 4 no stores (results wrong)
  uint32_t a_bytes, b_bytes;         // extents of the operands' plane images, from A / B
};

#define X_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define X_BARRIER() __builtin_amdgcn_s_barrier()

__device__ __forceinline__ void x_glds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

template <bool AKR, bool BKR, int EPI, int DIAG = 0>
__global__ __launch_bounds__(512, 1) void gemm_x3_dma_kernel(const X3Args g) {
  extern __shared__ __attribute__((aligned(16))) char x_lds[];
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
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)x_lds;

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
  auto zero_frag = [] { bf16x8 z; __builtin_memset(&z, 0, sizeof z); return z; };
  auto readA = [&](bf16x8 (&a)[4][3], const int slot) {
    const char* u = x_lds + slot * X_UNIT;
    int y = krY & 3;
    if (AKR) asm volatile("" : "+v"(y));        // keeps the four XORs inside the loop (see above)
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        if (DIAG == 2) a[f][p] = zero_frag();
        else if (!AKR) a[f][p] = *reinterpret_cast<const bf16x8*>(u + kcA + f * 3072 + p * 64);
        else a[f][p] = frag_kr(u + krBaseA + ((f ^ y) << 5), p);
      }
  };
  auto readB = [&](bf16x8 (&b)[2][3], const int slot) {
    const char* u = x_lds + slot * X_UNIT;
    int y = krY & 1;
    if (BKR) asm volatile("" : "+v"(y));
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        if (DIAG == 2) b[f][p] = zero_frag();
        else if (!BKR) b[f][p] = *reinterpret_cast<const bf16x8*>(u + kcB + f * 3072 + p * 64);
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
    if (DIAG == 3) return;
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
      f32x4* red = reinterpret_cast<f32x4*>(x_lds + X_RED);
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
    if (DIAG != 1) stage(true, 0, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bl, 0, 0);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 1: A-lo x B-hi
    readB(bo, rs); rs = next(rs);
    X_FENCE();
    if (DIAG != 1) stage(false, 1, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bo, 0, 2);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 2: A-hi x B-hi
    readA(a, rs); rs = next(rs);
    X_FENCE();
    if (DIAG != 1) stage(true, 1, t + 1, ws);
    ws = next(ws);
    X_WAIT_VM(9); X_BARRIER(); X_FENCE();
    mfma_phase(a, bo, 4, 2);
    X_FENCE(); X_BARRIER(); X_FENCE();
    // phase 3: A-hi x B-lo; B-lo(t + 1) into the registers B-hi(t) has left
    readB(bo, rs); rs = next(rs);
    X_FENCE();
    if (DIAG != 1) stage(false, 0, t + 2, ws);
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
  if (DIAG == 4) {
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) t += acc[i][j];
    if (t.x + t.y + t.z + t.w == 12345.678f) g.C[0] = t.x;
    return;
  }
  constexpr int EP_LD = 272;
  char* blk = x_lds + wave * 20480;
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
            mh[e] = in ? *reinterpret_cast<const s16x4*>(g.mask3 + (int64_t)row * ldm6 + i32_off(col)) : s16x4{0, 0, 0, 0};
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
          if (g.act == X_ACT_RELU) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        }
        *reinterpret_cast<f32x4*>(blk + (tm * 16 + c) * EP_LD + (tn * 16 + 4 * q) * 4) = v;
      }
    // the wave's own block: no barrier, LDS operations of one wave execute in order
    if constexpr (EPI == X_EPI_DW) {
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
          split_bf16x3(v0, a1, a2, a3); split_bf16x3(v1, b1, b2, b3);
          const uint4 o = pl == 0 ? make_uint4(a1.x, a1.y, b1.x, b1.y) : (pl == 1 ? make_uint4(a2.x, a2.y, b2.x, b2.y) : make_uint4(a3.x, a3.y, b3.x, b3.y));
          const int row = rowb + r;
          char* dst = c3w + (int64_t)row * ldc6 + ch * 16;
          if (g.diag == 5) dst = g.C3 + ((int64_t)(tile * 8 + wave) * 2 + half) * 24576 + it * 1024 + lane * 16;      // lab: 1 KB contiguous per instruction
          if (g.diag == 7) dst = g.C3 + ((int64_t)((blockIdx.x & 255) * 8 + wave)) * 4096 + (it & 3) * 1024 + lane * 16;   // lab: a 4 KB window per wave (stays in L2)
          if (g.diag == 8) { if (o.x == 0x12345678u && o.y == 0x9abcdef0u) *reinterpret_cast<uint4*>(dst) = o; continue; }   // lab: no plane stores
          if (g.diag == 6 && it % 3) continue;                                                                             // lab: a third of the plane stores
          if (row < g.M && colw + c8 < g.N) *reinterpret_cast<uint4*>(dst) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------------------
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3_host(float x, uint16_t p[3]) { p[0] = f2bf(x); float r = x - bf2f(p[0]); p[1] = f2bf(r); r = r - bf2f(p[1]); p[2] = f2bf(r); }

static float time_it(const std::function<void()>& f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // warm-up until the clocks have settled: after any idle stretch (host-side checks included) the first ~60 launches run 10-15 % slower than the
  // steady state (a first version warmed up with 3 launches and chased that difference through three epilogue variants)
  { hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1)); CK(hipEventRecord(w0));
    float ms = 0.f;
    while (ms < 120.f) { for (int i = 0; i < 10; i++) f(); CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1)); CK(hipEventElapsedTime(&ms, w0, w1)); } }
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms * 1000.f / iters;
}

template <typename K> static void set_lds(K k) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS)); }

int main(int argc, char** argv) {
  const int Bt = argc > 1 ? atoi(argv[1]) : 32768;
  const int IN = argc > 2 ? atoi(argv[2]) : 1024, OUT = argc > 3 ? atoi(argv[3]) : 1024;
  const int diag = argc > 4 ? atoi(argv[4]) : 0;
  const int SPLIT = argc > 5 ? atoi(argv[5]) : 0;
  const int iters = argc > 6 ? atoi(argv[6]) : 20;
  printf("layer %d -> %d at batch %d, fp32 values as three bf16 planes (I32 image)\n", IN, OUT, Bt);
  if (IN % 32 || OUT % 32 || Bt % 32) { printf("dims must be multiples of 32 in this lab\n"); return 1; }
  std::vector<float> hx((size_t)Bt * IN), hw((size_t)OUT * IN), hdy((size_t)Bt * OUT), hb(OUT);
  uint64_t sd = 88172645463325252ull;
  auto rnd = [&] { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
  for (auto& v : hx) { v = rnd(); if (v < -0.5f) v = 0.f; }     // relu-like: a quarter zeros
  for (auto& v : hw) v = rnd() * 0.05f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hdy) v = rnd();
  float *x, *wt, *dy, *bias, *y, *dx, *dw, *db;
  char *x3, *w3, *dy3, *y3, *dx3;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&wt, hw.size() * 4)); CK(hipMalloc(&dy, hdy.size() * 4)); CK(hipMalloc(&bias, OUT * 4));
  CK(hipMalloc(&y, (size_t)Bt * OUT * 4)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, OUT * 4));
  CK(hipMalloc(&x3, hx.size() * 6)); CK(hipMalloc(&w3, hw.size() * 6)); CK(hipMalloc(&dy3, hdy.size() * 6));
  CK(hipMalloc(&y3, (size_t)Bt * OUT * 6)); CK(hipMalloc(&dx3, hx.size() * 6));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), OUT * 4, hipMemcpyHostToDevice));
  auto split = [&](char* d, const float* s, int64_t rows, int cols) { hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, d, s, rows, cols, (int64_t)cols); };
  split(x3, x, Bt, IN); split(w3, wt, OUT, IN); split(dy3, dy, Bt, OUT);
  CK(hipDeviceSynchronize());
  {
    const float us = time_it([&] { split(x3, x, Bt, IN); }, 10);
    printf("split3 of x (%d x %d): %.1f us = %.2f TB/s (4 B read + 6 B written per element)\n", Bt, IN, us, (double)Bt * IN * 10 / us / 1e6);
    std::vector<char> h3(hx.size() * 6); CK(hipMemcpy(h3.data(), x3, h3.size(), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 20000; t++) {
      const size_t m = ((uint64_t)t * 2654435761u) % Bt, k = ((uint64_t)t * 40503u + 7) % IN;
      uint16_t p[3]; split3_host(hx[m * IN + k], p);
      for (int pl = 0; pl < 3; pl++) { uint16_t got; memcpy(&got, h3.data() + m * IN * 6 + (k >> 5) * 192 + pl * 64 + (k & 31) * 2, 2); if (got != p[pl]) bad++; }
    }
    printf("   split3 check: %d mismatches of 60000\n", bad);
  }
  auto k0 = gemm_x3_dma_kernel<false, false, X_EPI_FWD>; auto k1 = gemm_x3_dma_kernel<false, true, X_EPI_DX>; auto k2 = gemm_x3_dma_kernel<true, true, X_EPI_DW>;
  set_lds(k0); set_lds(k1); set_lds(k2);
  const double fl = 2.0 * Bt * IN * OUT;
  auto tiles = [](int M, int N) { return ((M + X_BM - 1) / X_BM) * ((N + X_BN - 1) / X_BN); };
  auto report = [&](const char* what, float us) { printf("%-44s %9.1f us  %7.1f TF/s fp32-equivalent (%.3f of 416.7)\n", what, us, fl / us / 1e6, fl / us / 1e6 / 416.7); fflush(stdout); };

  // forward
  X3Args a{}; a.A = x3; a.B = w3; a.C = y; a.C3 = y3; a.bias = bias; a.M = Bt; a.N = OUT; a.K = IN; a.lda = IN; a.ldb = IN; a.ldc = OUT; a.act = X_ACT_RELU; a.splitk = 1;
  a.a_bytes = (uint32_t)(hx.size() * 6); a.b_bytes = (uint32_t)(hw.size() * 6);
  report("fwd  (kc,kc) fp32 + planes out", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a); }, iters));
  {
    std::vector<float> hy((size_t)Bt * OUT); CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    std::vector<char> h3((size_t)Bt * OUT * 6); CK(hipMemcpy(h3.data(), y3, h3.size(), hipMemcpyDeviceToHost));
    double worst = 0; int bad3 = 0;
    for (int t = 0; t < 6000; t++) {
      const size_t m = ((uint64_t)t * 2654435761u) % Bt, n = ((uint64_t)t * 40503u + 7) % OUT;
      double s = hb[n], mass = fabs(hb[n]);
      for (int k = 0; k < IN; k++) { const double p = (double)hx[m * IN + k] * hw[n * IN + k]; s += p; mass += fabs(p); }
      if (s < 0) s = 0;
      worst = fmax(worst, fabs(hy[m * OUT + n] - s) / (mass + 1e-30));
      uint16_t p[3]; split3_host(hy[m * OUT + n], p);
      for (int pl = 0; pl < 3; pl++) { uint16_t got; memcpy(&got, h3.data() + m * OUT * 6 + (n >> 5) * 192 + pl * 64 + (n & 31) * 2, 2); if (got != p[pl]) bad3++; }
    }
    printf("   fwd check: worst |err| / term mass over 6000 samples = %.3e %s, plane mismatches %d\n", worst, worst < 2e-6 ? "ok" : "WRONG", bad3);
  }
  if (diag == 9) {      // time series: does the rate depend on how long the chip has been busy?
    auto t20 = [&] { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0));
      for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 50.f; };
    printf("series (us per launch, 20 launches each):"); for (int r = 0; r < 40; r++) printf(" %.0f", t20()); printf("\n");
    struct timespec ts = {0, 500000000}; nanosleep(&ts, nullptr);
    printf("after 0.5 s idle:"); for (int r = 0; r < 20; r++) printf(" %.0f", t20()); printf("\n");
    X3Args a8 = a; a8.diag = 8;
    for (int i = 0; i < 23; i++) hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a8);
    printf("after 23 launches without plane stores:"); for (int r = 0; r < 10; r++) printf(" %.0f", t20()); printf("\n");
    CK(hipMemset(y3, 0, (size_t)Bt * OUT * 6)); CK(hipDeviceSynchronize());
    printf("after a memset of the planes:"); for (int r = 0; r < 10; r++) printf(" %.0f", t20()); printf("\n");
    return 0;
  }
  if (diag) {
#define DIAGRUN(D, NAME) { auto kd = gemm_x3_dma_kernel<false, false, X_EPI_FWD, D>; set_lds(kd); \
      report("fwd  " NAME, time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a); }, iters)); }
    DIAGRUN(1, "no DMA in the loop (wrong)")
    DIAGRUN(2, "no fragment reads (wrong)")
    DIAGRUN(3, "no MFMAs (wrong)")
    DIAGRUN(4, "no epilogue (wrong)")
    for (int dg : {5, 6, 7, 8}) { X3Args a5 = a; a5.diag = dg; char nm[64]; snprintf(nm, sizeof nm, "fwd  plane-store variant %d (wrong)", dg);
      report(nm, time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a5); }, iters)); }
    report("fwd  again", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a); }, iters));
    X3Args a2 = a; a2.C3 = nullptr;
    report("fwd  fp32 output only", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a2); }, iters));
    hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), X_LDS, 0, a);
  }
  // dX = dy W, masked by relu'(x) read from plane 1 of x
  X3Args b{}; b.A = dy3; b.B = w3; b.C = dx; b.C3 = dx3; b.M = Bt; b.N = IN; b.K = OUT; b.lda = OUT; b.ldb = IN; b.ldc = IN; b.splitk = 1;
  b.mask = x; b.mask3 = x3; b.ldmask = IN;
  b.a_bytes = (uint32_t)(hdy.size() * 6); b.b_bytes = (uint32_t)(hw.size() * 6);
  report("dX   (kc,kr) fp32 + planes out, relu mask", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(512), X_LDS, 0, b); }, iters));
  {
    std::vector<float> hd(hx.size()); CK(hipMemcpy(hd.data(), dx, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 6000; t++) {
      const size_t m = ((uint64_t)t * 2654435761u) % Bt, n = ((uint64_t)t * 40503u + 7) % IN;
      double s = 0, mass = 0;
      for (int k = 0; k < OUT; k++) { const double p = (double)hdy[m * OUT + k] * hw[(size_t)k * IN + n]; s += p; mass += fabs(p); }
      if (!(hx[m * IN + n] > 0)) s = 0;
      worst = fmax(worst, fabs(hd[m * IN + n] - s) / (mass + 1e-30));
    }
    printf("   dX  check: worst |err| / term mass over 6000 samples = %.3e %s\n", worst, worst < 2e-6 ? "ok" : "WRONG");
  }
  // dW += dy^T x, split over k
  X3Args cc{}; cc.A = dy3; cc.B = x3; cc.C = dw; cc.M = OUT; cc.N = IN; cc.K = Bt; cc.lda = OUT; cc.ldb = IN; cc.ldc = IN;
  cc.Af32 = dy; cc.db = db;
  cc.a_bytes = (uint32_t)(hdy.size() * 6); cc.b_bytes = (uint32_t)(hx.size() * 6);
  {
    const int nt = tiles(cc.M, cc.N);
    int best = 1; double bu = 0;
    for (int sp = 1; sp <= 64 && sp * 8 <= Bt / X_BK; sp++) { const int nb = nt * sp; const double u = (double)nb / (((nb + 255) / 256) * 256); if (nb >= 128 && u > bu + 0.02) { bu = u; best = sp; } }
    cc.splitk = SPLIT ? SPLIT : best;
    CK(hipMemset(dw, 0, hw.size() * 4)); CK(hipMemset(db, 0, OUT * 4));
    hipLaunchKernelGGL(k2, dim3(nt * cc.splitk), dim3(512), X_LDS, 0, cc);
    std::vector<float> hd(hw.size()), hdb(OUT); CK(hipMemcpy(hd.data(), dw, hd.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdb.data(), db, OUT * 4, hipMemcpyDeviceToHost));
    double worst = 0, worstb = 0;
    for (int t = 0; t < 400; t++) {
      const size_t m = ((uint64_t)t * 2654435761u) % OUT, n = ((uint64_t)t * 40503u + 7) % IN;
      double s = 0, mass = 0, sb = 0, mb = 0;
      for (int k = 0; k < Bt; k++) { const double p = (double)hdy[(size_t)k * OUT + m] * hx[(size_t)k * IN + n]; s += p; mass += fabs(p); sb += hdy[(size_t)k * OUT + m]; mb += fabs(hdy[(size_t)k * OUT + m]); }
      worst = fmax(worst, fabs(hd[m * IN + n] - s) / (mass + 1e-30));
      worstb = fmax(worstb, fabs(hdb[m] - sb) / (mb + 1e-30));
    }
    printf("   dW  check: worst |err| / term mass over 400 samples = %.3e %s, db %.3e %s   (%d tiles x %d k-slices)\n", worst, worst < 2e-6 ? "ok" : "WRONG",
           worstb, worstb < 2e-6 ? "ok" : "WRONG", nt, cc.splitk);
    char nm[64]; snprintf(nm, sizeof nm, "dW   (kr,kr) atomics + db, split %d", cc.splitk);
    report(nm, time_it([&] { hipLaunchKernelGGL(k2, dim3(nt * cc.splitk), dim3(512), X_LDS, 0, cc); }, iters));
  }
  return 0;
}
