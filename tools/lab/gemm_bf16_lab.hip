// gemm_bf16_lab.hip -- stand-alone lab for the tensor-op (bf16 operand, fp32 accumulate) GEMM main loop (not part of the product).
//
// 256 x 256 x 64 tiles, one workgroup of 8 waves per CU (two per SIMD), v_mfma_f32_16x16x32_bf16, operands global -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass), two 64 KB LDS buffers.
//   * The waves are 2 (row groups) x 4 (column groups); a wave owns rows {64g..64g+63} + {128+64g..} and columns {32c..32c+31} +
//     {128+32c..} of the tile: 8 x 4 accumulators of 16 x 16.  A k-tile is four PHASES of 16 MFMAs (one quadrant of the wave's
//     output x 64 k):  1: A-lo x B-lo   2: A-lo x B-hi   3: A-hi x B-hi   4: A-hi x B-lo, each
//         { fragment reads of the phase (4 B + 8 A / 4 B / 8 A / none);  2 LDS-DMA pieces of a later k-tile;  s_waitcnt vmcnt(8);
//           s_barrier;  16 MFMAs at raised priority;  s_barrier }
//   * The two row groups run HALF A PHASE APART (group 1 passes one extra barrier at the start): while one wave of a SIMD issues
//     its 16 MFMAs the other one issues its reads and DMA pieces and waits at the barrier, so the matrix pipe always has work.
//   * An operand's k-tile is two UNITS of 16 KB (lo: rows/columns 0..127 of the tile, hi: 128..255), each read in exactly one
//     phase (A-lo, B-lo in 1, B-hi in 2, A-hi in 3) and restaged two phases after that read at the earliest:
//         phase 1 stages B-hi(t+1), 2: A-hi(t+1), 3: B-lo(t+2), 4: A-lo(t+2);
//     every phase waits vmcnt(8) after issuing its two pieces: four units stay in flight, a unit is waited for 4 phases after its
//     issue and one phase (and a barrier of both groups) before its first read.
//   * LDS images are lane-linear (the DMA writes base + 16 * lane); the swizzles live on the SOURCE address and in the reads:
//       k-contiguous operand: 128 unit-rows x 128 B, 16-byte chunk j of row u at slot j ^ ((u >> 1) & 7): a fragment (16 rows x 32 k)
//         is one conflict-free ds_read_b128 per lane;
//       rows-are-k operand: 64 k-rows x 256 B, chunk j (8 columns) of k-row r at slot j ^ (((r & 3) << 2) | ((r >> 2) & 3)); a
//         fragment is two ds_read_b64_tr_b16 (4 k x 16 columns each, transposed on the way out).
//   form 0  fwd: C[m][n] = act(sum_k A[m][k] B[n][k] + bias[n])      A, B k-contiguous
//   form 1  dX : C[m][n] = sum_k A[m][k] B[k][n]                      A k-contiguous, B rows-are-k
//   form 2  dW : C[m][n] (+)= sum_k A[k][m] B[k][n]                   A, B rows-are-k; split over k with atomics
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_bf16_lab.hip -o tools/lab/gemm_bf16_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int UNIT = 16384, BUF = 4 * UNIT;
constexpr int U_ALO = 0, U_AHI = 1, U_BLO = 2, U_BHI = 3;

struct Args {
  const uint16_t* A; const uint16_t* B; float* C; uint16_t* C16; const float* bias; const float* mask; int64_t ldmask;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int relu, atomic, splitk, eflags;     // eflags (experiments): 1 no fp32 store, 2 no bf16 store, 4 nontemporal stores
  unsigned a_bytes, b_bytes;
};

#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void glds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

// DIAG (timing only, results wrong): 1 no DMA in the loop, 2 no fragment reads, 3 no MFMAs
template <bool AKR, bool BKR, int DIAG = 0, int EPIV = 0>
__global__ __launch_bounds__(512, 1) void gemm16(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int c = lane & 15, q = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // ---- tile of this workgroup (n fastest; the workgroups of one XCD take neighbouring tiles) and its k range ----
  const unsigned nbx = (unsigned)((g.N + BN - 1) / BN), nby = (unsigned)((g.M + BM - 1) / BM), ntiles = nbx * nby;
  const unsigned total = gridDim.x, w = blockIdx.x;
  const unsigned xcd = w & 7u, loc = w >> 3, qq = total >> 3, rem = total & 7u;
  const unsigned nlin = xcd * qq + (xcd < rem ? xcd : rem) + loc;
  const unsigned tile = nlin % ntiles, ks = nlin / ntiles;          // ks: k-slice (splitk > 1)
  const unsigned by = tile / nbx, bx = tile - by * nbx;
  const int m0 = (int)by * BM, n0 = (int)bx * BN;
  const int nk_all = g.K / BK;
  const int kt0 = (int)((int64_t)nk_all * ks / g.splitk), kt1 = (int)((int64_t)nk_all * (ks + 1) / g.splitk);
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.B), 0, g.b_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;

  // ---- staging roles: the per-lane part of the source offset (bytes); piece i and the unit's position go into the scalar offset ----
  unsigned voffA, voffB;
  {
    const int u = tid >> 3, j = (tid & 7) ^ ((u >> 1) & 7);                                     // k-contiguous: unit-row, source chunk
    const int kr = tid >> 4, jr = (tid & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));             // rows-are-k: k-row, source chunk
    voffA = AKR ? (unsigned)((kr * g.lda + jr * 8) * 2) : (unsigned)((u * g.lda + j * 8) * 2);
    voffB = BKR ? (unsigned)((kr * g.ldb + jr * 8) * 2) : (unsigned)((u * g.ldb + j * 8) * 2);
  }
  const unsigned istepA = (unsigned)((AKR ? 32 : 64) * g.lda * 2), istepB = (unsigned)((BKR ? 32 : 64) * g.ldb * 2);
  auto stage = [&](const int unit, const int kt) {          // kt: k-tile index relative to kt0; beyond the range: loads that return 0 without touching memory
    const bool isA = unit < 2;
    const int hi = unit & 1;
    const int64_t ld = isA ? g.lda : g.ldb;
    const int o0 = (isA ? m0 : n0) + hi * 128;
    const bool kr = isA ? AKR : BKR;
    unsigned soff;
    if (kt >= nk) soff = isA ? g.a_bytes : g.b_bytes;
    else soff = kr ? (unsigned)(((int64_t)(kt0 + kt) * BK * ld + o0) * 2) : (unsigned)(((int64_t)o0 * ld + (int64_t)(kt0 + kt) * BK) * 2);
    const unsigned dst = lds_base + (unsigned)((kt & 1) * BUF + unit * UNIT) + (unsigned)wave * 1024u;
    if (DIAG == 1 && kt >= 2) return;
    glds16(isA ? voffA : voffB, isA ? rsA : rsB, dst, soff);
    glds16(isA ? voffA : voffB, isA ? rsA : rsB, dst + 8192u, soff + (isA ? istepA : istepB));
  };

  // ---- fragment read offsets (bytes inside a unit) ----
  // k-contiguous: lane (c, q) reads chunk 4s + q of unit-row base + 16 f + c; the swizzle term is (c >> 1) for every fragment
  const int kcA0 = (grp * 64 + c) * 128 + ((q ^ (c >> 1)) << 4), kcA1 = (grp * 64 + c) * 128 + (((4 + q) ^ (c >> 1)) << 4);
  const int kcB0 = (wc * 32 + c) * 128 + ((q ^ (c >> 1)) << 4), kcB1 = (wc * 32 + c) * 128 + (((4 + q) ^ (c >> 1)) << 4);
  // rows-are-k: lane (q; tq, tp) reads 8 bytes at k-row 32 s + 8 q + tq (+ 4), chunk (o >> 3) + (tp >> 1), o = first row/column of the fragment
  const int krX1 = (tq << 2) | (2 * (q & 1)), krX2 = krX1 | 1;
  const int krRow = (8 * q + tq) * 256 + 8 * (tp & 1);
  auto fragA = [&](const char* unit, int f, int s) -> bf16x8 {
    if (DIAG == 2) return bf16x8{};
    if (!AKR) return *reinterpret_cast<const bf16x8*>(unit + (s ? kcA1 : kcA0) + f * 2048);
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const int ch = ((grp * 64 + f * 16) >> 3) + (tp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unit + s * 8192 + krRow + ((ch ^ krX1) << 4)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unit + s * 8192 + krRow + 1024 + ((ch ^ krX2) << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto fragB = [&](const char* unit, int f, int s) -> bf16x8 {
    if (DIAG == 2) return bf16x8{};
    if (!BKR) return *reinterpret_cast<const bf16x8*>(unit + (s ? kcB1 : kcB0) + f * 2048);
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const int ch = ((wc * 32 + f * 16) >> 3) + (tp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unit + s * 8192 + krRow + ((ch ^ krX1) << 4)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unit + s * 8192 + krRow + 1024 + ((ch ^ krX2) << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  bf16x8 aLo[4][2], aHi[4][2], bLo[2][2], bHi[2][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mfma_block = [&](bf16x8 (&a)[4][2], bf16x8 (&b)[2][2], const int tm0, const int tn0) {
    if (DIAG == 3) {       // keep the fragments alive, issue nothing
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int s = 0; s < 2; s++) asm volatile("" ::"v"(a[tm][s]));
#pragma unroll
      for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int s = 0; s < 2; s++) asm volatile("" ::"v"(b[tn][s]));
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int s = 0; s < 2; s++)
          acc[tm0 + tm][tn0 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn][s], a[tm][s], acc[tm0 + tm][tn0 + tn], 0, 0, 0);   // operands swapped: a lane holds 4 consecutive columns
    __builtin_amdgcn_s_setprio(0);
  };
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIER() __builtin_amdgcn_s_barrier()

  // ---- prologue: k-tile 0 and the first two units of k-tile 1 ----
  stage(U_BLO, 0); stage(U_ALO, 0); stage(U_BHI, 0); stage(U_AHI, 0); stage(U_BLO, 1); stage(U_ALO, 1);
  WAIT_VM(8);          // B-lo(0), A-lo(0) have landed
  BARRIER();
  FENCE();
  if (grp == 1) BARRIER();      // group 1 runs half a phase behind group 0
  FENCE();

  for (int t = 0; t < nk; t++) {
    const char* cb = lds + (t & 1) * BUF;
    // phase 1: A-lo x B-lo
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) bLo[f][s] = fragB(cb + U_BLO * UNIT, f, s);
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) aLo[f][s] = fragA(cb + U_ALO * UNIT, f, s);
    FENCE();
    stage(U_BHI, t + 1);
    WAIT_VM(8);
    BARRIER();
    FENCE();
    mfma_block(aLo, bLo, 0, 0);
    FENCE();
    BARRIER();
    FENCE();
    // phase 2: A-lo x B-hi
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) bHi[f][s] = fragB(cb + U_BHI * UNIT, f, s);
    FENCE();
    stage(U_AHI, t + 1);
    WAIT_VM(8);
    BARRIER();
    FENCE();
    mfma_block(aLo, bHi, 0, 2);
    FENCE();
    BARRIER();
    FENCE();
    // phase 3: A-hi x B-hi
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int s = 0; s < 2; s++) aHi[f][s] = fragA(cb + U_AHI * UNIT, f, s);
    FENCE();
    stage(U_BLO, t + 2);
    WAIT_VM(8);
    BARRIER();
    FENCE();
    mfma_block(aHi, bHi, 4, 2);
    FENCE();
    BARRIER();
    FENCE();
    // phase 4: A-hi x B-lo
    stage(U_ALO, t + 2);
    WAIT_VM(8);
    BARRIER();
    FENCE();
    mfma_block(aHi, bLo, 4, 0);
    FENCE();
    BARRIER();
    FENCE();
  }
  if (grp == 0) BARRIER();
  WAIT_VM(0);

  // ---- epilogue: lane (c, q) holds C[row(tm) + c][col(tn) + 4 q + {0..3}] ----
  if (DIAG == 4) {
#pragma unroll
    for (int tm = 0; tm < 8; tm++)
#pragma unroll
      for (int tn = 0; tn < 4; tn++) asm volatile("" ::"v"(acc[tm][tn]));
    return;
  }
  if (g.atomic) {
    // partial tile (k-slice): added to C by atomics, through the per-wave LDS block so that one instruction covers 2 rows x 128
    // contiguous bytes (4-byte pieces 16 bytes apart run ~10x slower at the memory-side adders)
    char* blk = lds + 2 * BUF + wave * 4096;
    const int rr = lane >> 5, rc = lane & 31;
#pragma unroll
    for (int tm = 0; tm < 8; tm++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int row0 = m0 + (tm >= 4 ? 128 : 0) + grp * 64 + (tm & 3) * 16;
        const int col = n0 + h * 128 + wc * 32 + rc;
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) *reinterpret_cast<f32x4*>(blk + c * 144 + (t2 * 16 + 4 * q) * 4) = acc[tm][2 * h + t2];
#pragma unroll
        for (int p = 0; p < 8; p++) {
          const float v = *reinterpret_cast<const float*>(blk + (2 * p + rr) * 144 + rc * 4);
          const int row = row0 + 2 * p + rr;
          if (row < g.M && col < g.N) atomicAdd(g.C + (int64_t)row * g.ldc + col, v);
        }
      }
    return;
  }
  if (EPIV == 1) {
    // through a per-wave LDS block (16 rows x 32 columns, rows padded to 144 B) so that a store instruction covers 8 rows x 128
    // contiguous bytes (whole lines) instead of 16 rows x 64 B
    char* blk = lds + 2 * BUF + wave * 4096;
    const int rr = lane >> 3, rc = lane & 7;
#pragma unroll
    for (int tm = 0; tm < 8; tm++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int row0 = m0 + (tm >= 4 ? 128 : 0) + grp * 64 + (tm & 3) * 16;
        const int col0 = n0 + h * 128 + wc * 32;
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
          f32x4 v = acc[tm][2 * h + t2];
          if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col0 + t2 * 16 + 4 * q);
          if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          *reinterpret_cast<f32x4*>(blk + c * 144 + (t2 * 16 + 4 * q) * 4) = v;
        }
        // the wave's own block: no barrier, the compiler orders the LDS accesses
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(blk + (p * 8 + rr) * 144 + rc * 16);
          const int row = row0 + p * 8 + rr, col = col0 + rc * 4;
          if (row < g.M && col < g.N && g.eflags) {     // experiment variants of the store pair
            f32x4 v = vv;
            float* cp = g.C + (int64_t)row * g.ldc + col;
            const bf16x4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            if (g.eflags & 4) {
              if (!(g.eflags & 1)) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cp));
              if (!(g.eflags & 2) && g.C16) __builtin_nontemporal_store(t, reinterpret_cast<bf16x4*>(g.C16 + (int64_t)row * g.ldc + col));
            } else {
              if (!(g.eflags & 1)) *reinterpret_cast<f32x4*>(cp) = v;
              if (!(g.eflags & 2) && g.C16) *reinterpret_cast<bf16x4*>(g.C16 + (int64_t)row * g.ldc + col) = t;
            }
          } else if (row < g.M && col < g.N) {
            f32x4 v = vv;
            if (g.mask) {
              const f32x4 mk = *reinterpret_cast<const f32x4*>(g.mask + (int64_t)row * g.ldmask + col);
              v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<f32x4*>(g.C + (int64_t)row * g.ldc + col) = v;
            if (g.C16) {
              const bf16x4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
              *reinterpret_cast<bf16x4*>(g.C16 + (int64_t)row * g.ldc + col) = t;
            }
          }
        }
      }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < 8; tm++) {
    const int row = m0 + (tm >= 4 ? 128 : 0) + grp * 64 + (tm & 3) * 16 + c;
    if (row >= g.M) continue;
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
      const int col = n0 + (tn >= 2 ? 128 : 0) + wc * 32 + (tn & 1) * 16 + 4 * q;
      if (col >= g.N) continue;
      f32x4 v = acc[tm][tn];
      float* cp = g.C + (int64_t)row * g.ldc + col;
      if (g.atomic) {
        atomicAdd(cp + 0, v.x); atomicAdd(cp + 1, v.y); atomicAdd(cp + 2, v.z); atomicAdd(cp + 3, v.w);
        continue;
      }
      if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
      if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<f32x4*>(cp) = v;
      if (g.C16) {
        const bf16x4 t = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4*>(g.C16 + (int64_t)row * g.ldc + col) = t;
      }
    }
  }
}

static float time_it(const std::function<void()>& f, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 30; i++) f();          // the clock needs tens of ms to ramp
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1000.0f / iters;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int Bt = argc > 1 ? atoi(argv[1]) : 32768;
  const int IN = argc > 2 ? atoi(argv[2]) : 1024, OUT = argc > 3 ? atoi(argv[3]) : 1024;
  const int diag = argc > 4 ? atoi(argv[4]) : 0;
  const int SPLIT = argc > 5 ? atoi(argv[5]) : 0;
  printf("layer %d -> %d at batch %d, bf16 operands\n", IN, OUT, Bt);
  std::vector<uint16_t> hx((size_t)Bt * IN), hw((size_t)OUT * IN), hdy((size_t)Bt * OUT);
  std::vector<float> hb(OUT);
  uint64_t sd = 88172645463325252ull;
  auto rnd = [&] { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
  for (auto& v : hx) v = f2bf(rnd());
  for (auto& v : hw) v = f2bf(rnd() * 0.05f);
  for (auto& v : hb) v = rnd();
  for (auto& v : hdy) v = f2bf(rnd());
  uint16_t *x, *wt, *dy, *y16; float *bias, *y, *dx, *dw;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&wt, hw.size() * 2)); CK(hipMalloc(&dy, hdy.size() * 2)); CK(hipMalloc(&bias, OUT * 4));
  CK(hipMalloc(&y, (size_t)Bt * OUT * 4)); CK(hipMalloc(&y16, (size_t)Bt * (OUT > IN ? OUT : IN) * 2)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wt, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), OUT * 4, hipMemcpyHostToDevice));
  auto k0 = gemm16<false, false>; auto k1 = gemm16<false, true, 0, 1>; auto k2 = gemm16<true, true>;
  CK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
  CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
  CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
  const double fl = 2.0 * Bt * IN * OUT;
  auto tiles = [](int M, int N) { return ((M + BM - 1) / BM) * ((N + BN - 1) / BN); };
  auto report = [&](const char* what, float us) { printf("%-40s %9.1f us  %7.1f TF/s\n", what, us, fl / us / 1e6); };

  // forward
  Args a{}; a.A = x; a.B = wt; a.C = y; a.C16 = y16; a.bias = bias; a.M = Bt; a.N = OUT; a.K = IN; a.lda = IN; a.ldb = IN; a.ldc = OUT; a.relu = 1; a.splitk = 1;
  a.a_bytes = (unsigned)(hx.size() * 2); a.b_bytes = (unsigned)(hw.size() * 2);
  report("fwd  (kc,kc) fp32 + bf16 outputs", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a); }, 30));
  if (diag == 2) {
    auto kd = gemm16<false, false, 0, 1>; CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
    auto k4 = gemm16<false, false, 4, 0>; CK(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
    report("no stores at all", time_it([&] { hipLaunchKernelGGL(k4, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a); }, 30));
    for (int ef : {0, 1, 2, 3, 4, 5, 6}) {
      Args ae = a; ae.eflags = ef;
      char nm[64]; snprintf(nm, sizeof nm, "via LDS, eflags %d", ef);
      report(nm, time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, ae); }, 30));
    }
    return 0;
  }
  if (diag) {
#define DIAGRUN(D, NAME) { auto kd = gemm16<false, false, D>; CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768)); \
      report("fwd  " NAME, time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a); }, 30)); }
    DIAGRUN(1, "no DMA in the loop (wrong)")
    DIAGRUN(2, "no fragment reads (wrong)")
    DIAGRUN(3, "no MFMAs (wrong)")
    DIAGRUN(4, "no stores (wrong)")
    { auto kd = gemm16<false, false, 0, 1>; CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768));
      report("fwd  whole-line stores via LDS", time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a); }, 30));
      Args a3 = a; a3.K = 64;
      report("fwd  K = 64 only (TF/s meaningless)", time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a3); }, 30));
      report("fwd  K = 64 only, direct stores", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a3); }, 30));
      hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a); }
    Args a2 = a; a2.C16 = nullptr;
    report("fwd  fp32 output only", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(512), 2 * BUF + 32768, 0, a2); }, 30));
  }
  {
    std::vector<float> hy((size_t)Bt * OUT); CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint16_t> h16((size_t)Bt * OUT); CK(hipMemcpy(h16.data(), y16, h16.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; int bad16 = 0;
    for (int t = 0; t < 6000; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % Bt), n = (int)(((uint64_t)t * 40503u + 7) % OUT);
      double s = hb[n], mass = fabs(hb[n]);
      for (int k = 0; k < IN; k++) { const double p = (double)bf2f(hx[(size_t)m * IN + k]) * bf2f(hw[(size_t)n * IN + k]); s += p; mass += fabs(p); }
      if (s < 0) s = 0;
      worst = fmax(worst, fabs(hy[(size_t)m * OUT + n] - s) / (mass + 1e-30));
      if (h16[(size_t)m * OUT + n] != f2bf(hy[(size_t)m * OUT + n])) bad16++;
    }
    printf("   fwd check: worst |err| / term mass over 6000 samples = %.3e %s, twin mismatches %d\n", worst, worst < 1e-5 ? "ok" : "WRONG", bad16);
  }
  // dX = dy W
  Args b{}; b.A = dy; b.B = wt; b.C = dx; b.C16 = y16; b.M = Bt; b.N = IN; b.K = OUT; b.lda = OUT; b.ldb = IN; b.ldc = IN; b.splitk = 1;
  b.a_bytes = (unsigned)(hdy.size() * 2); b.b_bytes = (unsigned)(hw.size() * 2);
  report("dX   (kc,kr) fp32 + bf16 outputs", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(512), 2 * BUF + 32768, 0, b); }, 30));
  { Args b2 = b; b2.C16 = nullptr;
    report("dX   fp32 output only", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(512), 2 * BUF + 32768, 0, b2); }, 30));
    float* xm; CK(hipMalloc(&xm, hx.size() * 4)); CK(hipMemset(xm, 0x3f, hx.size() * 4));    // all positive: the mask passes everything
    Args b3 = b; b3.mask = xm; b3.ldmask = IN;
    report("dX   fp32 + bf16 outputs, relu mask", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(512), 2 * BUF + 32768, 0, b3); }, 30));
    CK(hipFree(xm)); }
  {
    std::vector<float> hd(hx.size()); CK(hipMemcpy(hd.data(), dx, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 6000; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % Bt), n = (int)(((uint64_t)t * 40503u + 7) % IN);
      double s = 0, mass = 0;
      for (int k = 0; k < OUT; k++) { const double p = (double)bf2f(hdy[(size_t)m * OUT + k]) * bf2f(hw[(size_t)k * IN + n]); s += p; mass += fabs(p); }
      worst = fmax(worst, fabs(hd[(size_t)m * IN + n] - s) / (mass + 1e-30));
    }
    printf("   dX  check: worst |err| / term mass over 6000 samples = %.3e %s\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  }
  // dW += dy^T x, split over k
  Args cc{}; cc.A = dy; cc.B = x; cc.C = dw; cc.M = OUT; cc.N = IN; cc.K = Bt; cc.lda = OUT; cc.ldb = IN; cc.ldc = IN; cc.atomic = 1;
  cc.a_bytes = (unsigned)(hdy.size() * 2); cc.b_bytes = (unsigned)(hx.size() * 2);
  {
    const int nt = tiles(cc.M, cc.N);
    int best = 1; double bu = 0;
    for (int sp = 1; sp <= 64 && sp * 4 <= Bt / BK; sp++) { const int nb = nt * sp; const double u = (double)nb / (((nb + 255) / 256) * 256); if (nb >= 256 && u > bu + 0.02) { bu = u; best = sp; } }
    cc.splitk = SPLIT ? SPLIT : best;
    CK(hipMemset(dw, 0, hw.size() * 4));
    hipLaunchKernelGGL(k2, dim3(nt * cc.splitk), dim3(512), 2 * BUF + 32768, 0, cc);
    std::vector<float> hd(hw.size()); CK(hipMemcpy(hd.data(), dw, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 400; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % OUT), n = (int)(((uint64_t)t * 40503u + 7) % IN);
      double s = 0, mass = 0;
      for (int k = 0; k < Bt; k++) { const double p = (double)bf2f(hdy[(size_t)k * OUT + m]) * bf2f(hx[(size_t)k * IN + n]); s += p; mass += fabs(p); }
      worst = fmax(worst, fabs(hd[(size_t)m * IN + n] - s) / (mass + 1e-30));
    }
    printf("   dW  check: worst |err| / term mass over 400 samples = %.3e %s   (%d tiles x %d k-slices)\n", worst, worst < 1e-5 ? "ok" : "WRONG", nt, cc.splitk);
    char nm[64]; snprintf(nm, sizeof nm, "dW   (kr,kr) atomics, split %d", cc.splitk);
    report(nm, time_it([&] { hipLaunchKernelGGL(k2, dim3(nt * cc.splitk), dim3(512), 2 * BUF + 32768, 0, cc); }, 30));
  }
  return 0;
}
