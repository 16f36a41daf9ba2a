// linear_sk.hip -- the big fp32 GEMMs of Linear forward / backward (exact fp32, v_mfma_f32_16x16x4_f32): persistent workgroups,
// one per CU, whose matrix pipe never waits.
//
// The register-staged kernel of linear.hip (128 x 128 x 16 tiles, three workgroups per CU, the compiler's schedule) keeps the
// matrix pipe busy 0.84 of the time on the layers of the 32768-sample step; the shape of the loop below -- the one hand-written
// library GEMMs use -- keeps it busy 0.93-0.96:
//   * ONE workgroup per CU (4 waves, one per SIMD, 2 x 2 over a 128 x 128 tile, 64 accumulator registers in AGPRs), k-tiles of 64:
//     a wave issues 256 MFMAs (8192 matrix-pipe cycles) per k-tile back to back and EVERY other instruction sits in the shadow of
//     one of them, in a fixed order (sched_barrier after every MFMA):
//       MFMA   0..23   the 24 fragment reads (ds_read_b128) of k-groups 1..3 of this k-tile -- the whole k-tile's fragments live
//                      in 128 VGPRs, so the single LDS buffer is free again after ~50 MFMAs
//       MFMA  53       s_waitcnt lgkmcnt(0) + s_barrier: every wave has its fragments
//       MFMA  54..219  16 x { ds_write_b128 of the NEXT k-tile (in registers since the previous iteration);
//                             buffer_load_dwordx4 of the one after it into the same registers }, one pair per 11 MFMAs
//       MFMA 243       s_waitcnt lgkmcnt(0) + s_barrier: the next k-tile is in LDS
//       MFMA 244..251  the 8 fragment reads of its first k-group
//     (global loads have a whole iteration -- 3.4 us -- to land; no wait in the loop is ever for something issued recently);
//   * the operand stream is CONTINUOUS ACROSS OUTPUT TILES: a workgroup walks its tiles with the loads two k-tiles ahead of the
//     MFMAs, so a tile's first operands arrive while the previous tile's last MFMAs run; only the accumulator store sits between;
//   * work is dealt in whole tiles (forward, dX: tiles i*G + (w%8)*(G/8) + w/8 -- the workgroups of one XCD walk neighbouring
//     tiles, their A panels stay in that XCD's L2) or, for the weight gradient (few tiles, 32768-deep reduction, result
//     accumulated anyway), STREAM-K: the flat (tile, k-tile) space cut into G equal ranges, partial tiles added by atomics;
//   * LDS images: a k-contiguous operand keeps its rows (256 B + 32 B of padding per 1 KiB; fragment = one ds_read_b128 = four
//     k-steps of one 16-row tile: lane (c, q) of the 16x16x4 MFMA takes k = 16j + 4q + e, legal because both operands agree);
//     a rows-are-k operand (w in dX; dy and x in dW) is stored as it arrives and one ds_read_b128 yields ONE k-step of FOUR
//     16-row tiles (tile t holds rows 4c + t): both forms read 16 x 16 B per operand and k-tile, conflict-free, no transpose;
//   * the MFMAs are inline asm with the accumulator tied ("+a"): as builtins hipcc gives the loop-carried accumulators VGPR-class
//     phis and moves all 64 through v_accvgpr_read / _write at the head of every iteration.  The hazard recogniser does not see
//     inline-asm MFMAs: s_nop pads sit behind the loop (MFMA result -> v_accvgpr_read) and behind the zeroing.
// Arithmetic: an exact fp32 fmaf chain per output element, k visited in the order 16j + 4q + e (q = lane / 16): a fixed order,
// different from the other kernels' -- parity tests hold every kernel to 1e-5 of the term mass against the oracle.
// Serves: M % 128 == 0, N % 128 == 0, K % 64 == 0, 16-byte aligned operands, enough tiles to fill the chip evenly; everything
// else stays with linear.hip.  Developed in tools/lab/gemm_sk_lab.hip (stand-alone, with ablation modes).
//
// Replaces cublasSgemm of Linear::forward_kernel / backward_kernel [ref: src/ops/linear.cu:436-453,624-659].
#include "linear_gemm.h"

#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include <utility>

using namespace ffh_gemm;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_BM = 128, SK_BN = 128, SK_BK = 64;
constexpr int SK_LDS_KC = 128 * 256 + 32 * 32;   // k-contiguous operand: row r at r*256 + (r>>2)*32
constexpr int SK_LDS_KR = 64 * 512;              // rows-are-k operand: row k at k*512
// 64-row tiles (TMW = 2: a wave owns 32 x 64 of a 64 x 128 tile; round 5): the forward / data-gradient forms of layers whose 128-row
// tiles are fewer than the CUs -- 1024 -> 512 at 4096 samples is 128 tiles of 128 x 128 (stream-K with fix-up: 8 k-tiles per workgroup and a
// round trip of partial tiles through memory, 95 TFLOP/s) but 256 tiles of 64 x 128, one per CU with the whole reduction.  The A image
// pads 32 bytes every TMW rows (a wave's m-tile r holds rows TMW c + r: the fragment reads of lanes c = 0 .. 15 stay on sixteen bank groups).
constexpr int sk_lds_kc_a(int TMW) { return 32 * TMW * 256 + 32 * 32; }     // BM = 32 TMW rows: BM * 256 + (BM / TMW) * 32

enum { SK_EPI_FWD = 0, SK_EPI_DX_STORE = 1, SK_EPI_DX_ADD = 2, SK_EPI_DW_ATOMIC = 3, SK_EPI_DX_CMAP = 4 };

struct SkArgs {
  const float* A; const float* B; float* C;
  const float* bias;           // FWD: per-column bias or null
  float*       db;             // DW (DB instantiation): db[m] += sum_k A(k, m), the bias gradient [ref: src/ops/linear.cu:644-651]
  const float* mask;           // DX: C = mask[m][n] > 0 ? v : 0 (relu' of the layer below) or null
  const ffh_col_dest* colmap;  // DX_CMAP (ffh_linear_bwd_set_dx_scatter): column n of C lives at colmap[n].base[m * colmap[n].ld] -- the exchange
                               // path's first top layer stores its data gradient into the bottom MLP's gradient and the all-to-all send buffer
  float*       colsum;         // DX_STORE (ffh_linear_bwd_set_dx_colsum): colsum[n] += sum over the tile's rows of what is stored -- the bias
                               // gradient of the layer below, whose dy this C is -- or null
  int64_t lda, ldb, ldc, ldmask;
  int M, N, K;
  int act;
  unsigned a_bytes, b_bytes, bias_bytes, mask_bytes;     // extents for the buffer descriptors (0: operand absent -> loads return 0)
  // SPLIT (stream-K forward / data gradient): the partial tiles of a tile that straddles ranges -- slot 2r: range r's LEADING segment
  // (it continues a tile an earlier range began), slot 2r + 1: its TRAILING segment (the head of a tile later ranges finish), 128 x 128
  // accumulators in lane order -- and one arrival counter per range (the counter of the range holding the tile's head counts the
  // tile's parts): the part that arrives last adds them up, and leaves the counter at 0 (so a launch finds every counter 0, also
  // when it is a node of a replayed hipGraph: no per-launch argument)
  float*    sk_slots;
  unsigned* sk_cnt;
};

template <int... I, class F>
__device__ __forceinline__ void sk_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N_, class F>
__device__ __forceinline__ void sk_static_for(F&& f) { sk_static_for_impl(std::make_integer_sequence<int, N_>{}, static_cast<F&&>(f)); }

#define SK_PIN() __builtin_amdgcn_sched_barrier(0)

constexpr int SK_EP_LD = 68;                       // floats per row of a wave's epilogue image (DW): 64 + 4 of padding
constexpr int SK_EP_ROWS = 16;                     // rows of the wave's 64 x 64 result that pass through the image at a time (the rows the lanes of
                                                   // one q hold): 4 rounds.  With all 64 rows at once the kernel held 135 KB of LDS and the 33-37 KB
                                                   // workgroups of the small layers' GEMMs could not start beside it (at 4096 samples the bottom MLP's
                                                   // backward waited ~200 us of a 1.2 ms step for the big weight-gradient GEMM's workgroups to leave);
                                                   // a segment ends once or twice per workgroup, so the rounds cost nothing
constexpr int SK_EP_WAVE = SK_EP_ROWS * SK_EP_LD * 4;     // bytes per wave

// SPLIT: stream-K for the forms whose output is stored, not accumulated (forward, data gradient).  Whole tiles only fill the chip
// when their count is a multiple of the workgroup count: the 864 tiles of 3456 -> 1024's data gradient at 4096 samples are 3.4
// rounds of 256 (a quarter of the last round's CUs idle: 114 instead of 135 TFLOP/s), the 128 tiles of 1024 -> 512 leave half the
// chip empty.  As for the weight gradient the flat (tile, k-tile) space is cut into G equal ranges; a tile that straddles ranges
// is computed in parts, and NOBODY WAITS (round 5; round 4's owner spun on flags that later workgroups of the same launch set --
// forward progress then hung on every contributor becoming resident while the owners held their CUs, which nothing guarantees
// beside another persistent kernel or under a CU mask): every part leaves its accumulators in its slot (lane order: coalesced
// 8-byte agent-scope stores), then one lane adds 1 to the tile's arrival counter; the part whose add returns parts - 1 arrived
// last: it adds the parts up in k order -- a fixed order whoever is last: same bits run to run, so the form also runs in
// deterministic mode -- and runs the ordinary epilogue.  A range computes its leading segment first and its trailing one (the head
// of the next tile) last, so the head usually is the last to arrive and its own part never leaves its registers.  Cross-workgroup
// values travel as agent-scope relaxed atomics (sc1: written through / read behind the per-XCD L2s), ordered by completion (vmcnt
// + barrier before the counter add; the loads behind the returned add + a barrier), as in embedding.hip's folds.
template <bool AKR, bool BKR, int EPI, bool DB = false, bool SPLIT = false, int TMW = 4>
__global__ __launch_bounds__(256, 1) void gemm_sk_kernel(const SkArgs g) {
  static_assert(!DB || (EPI == SK_EPI_DW_ATOMIC && AKR), "the bias gradient rides on the weight-gradient form");
  static_assert(!SPLIT || EPI != SK_EPI_DW_ATOMIC, "the weight gradient meets by atomics");
  static_assert(TMW == 4 || (TMW == 2 && !AKR && EPI != SK_EPI_DW_ATOMIC), "64-row tiles: forward and data gradient (A k-contiguous)");
  extern __shared__ __attribute__((aligned(16))) char sk_lds[];
  constexpr bool ATOMIC = EPI == SK_EPI_DW_ATOMIC;
  constexpr bool STREAMK = ATOMIC || SPLIT;
  constexpr int BM = 32 * TMW;                      // rows of the tile: 128 or 64
  constexpr int NM = 64 * TMW;                      // MFMAs per k-tile and wave
  constexpr int NPA = 2 * TMW, NP = NPA + 8;        // staging pieces (float4 per thread) of a k-tile: A, A + B
  constexpr int LDS_A = AKR ? SK_LDS_KR : sk_lds_kc_a(TMW);
  char* const ldsA = sk_lds;
  char* const ldsB = sk_lds + LDS_A;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int c16 = lane & 15, q = lane >> 4;

  // fragment read addresses (bytes inside the operand's LDS image)
  const int fra = AKR ? (q * 2048 + wy * 256 + c16 * 16) : ((16 * TMW * wy + TMW * c16) * 256 + (16 * wy + c16) * 32 + q * 16);
  const int frb = BKR ? (q * 2048 + wx * 256 + c16 * 16) : ((64 * wx + 4 * c16) * 256 + (16 * wx + c16) * 32 + q * 16);
  // staging roles: this thread's 8 + 8 float4 of a k-tile -- where they come from (per-lane byte offset in the matrix) and go to
  const int srowA = AKR ? (tid >> 5) : (tid >> 4), schA = AKR ? (tid & 31) : (tid & 15);
  const int srowB = BKR ? (tid >> 5) : (tid >> 4), schB = BKR ? (tid & 31) : (tid & 15);
  const unsigned voffA = (unsigned)((srowA * g.lda + schA * 4) * 4);
  const unsigned voffB = (unsigned)((srowB * g.ldb + schB * 4) * 4);
  const int swA = AKR ? (srowA * 512 + schA * 16) : (srowA * 256 + (srowA / TMW) * 32 + schA * 16);
  const int swB = BKR ? (srowB * 512 + schB * 16) : (srowB * 256 + (srowB >> 2) * 32 + schB * 16);
  constexpr int SW_STEP_A = AKR ? 4096 : (4096 + (16 / TMW) * 32), SW_STEP_B = BKR ? 4096 : 4224;     // LDS bytes between a thread's consecutive pieces
  const unsigned ioffA = (unsigned)((AKR ? 8 : 16) * g.lda * 4), ioffB = (unsigned)((BKR ? 8 : 16) * g.ldb * 4);   // global bytes between them
  const unsigned kadvA = AKR ? (unsigned)(SK_BK * g.lda * 4) : (unsigned)(SK_BK * 4), kadvB = BKR ? (unsigned)(SK_BK * g.ldb * 4) : (unsigned)(SK_BK * 4);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);

  // ---- this workgroup's share of the (tile, k-tile) iteration space ----
  const unsigned nbx = (unsigned)(g.N / SK_BN), nby = (unsigned)(g.M / BM), ntiles = nbx * nby;
  const unsigned nk = (unsigned)(g.K / SK_BK);
  const unsigned G = gridDim.x, w = blockIdx.x;
  const unsigned total_it = ntiles * nk;                  // < 2^32 (the host checks)
  const unsigned wperm = (w & 7u) * (G >> 3) + (w >> 3);  // G is a multiple of 8
  unsigned it_b, it_e;
  const unsigned ipw = (total_it + G - 1) / G;
  const unsigned rng = SPLIT ? wperm : w;                 // SPLIT: the ranges of one XCD's workgroups are neighbours (shared panels in its L2)
  if (STREAMK) {
    it_b = rng * ipw; it_e = it_b + ipw < total_it ? it_b + ipw : total_it;
    if (it_b > total_it) it_b = total_it;
  } else {
    const unsigned mine = ntiles / G + (wperm < ntiles % G ? 1u : 0u);
    it_b = 0; it_e = mine * nk;
  }
  const unsigned n_it = it_e - it_b;
  if (n_it == 0) return;

  struct Cursor { unsigned seq, kt, m0, n0, offA, offB; };   // seq: index of the tile in this workgroup's sequence
  auto place = [&](Cursor& c) {      // tile coordinates and operand offsets of (c.seq, c.kt)
    unsigned lin;
    if (STREAMK) lin = it_b / nk + c.seq;
    else lin = c.seq * G + wperm;
    if (lin >= ntiles) lin = ntiles - 1;                  // run-ahead loads past the end of the share: any valid tile
    const unsigned by = lin / nbx, bx = lin - by * nbx;
    c.m0 = by * BM; c.n0 = bx * SK_BN;
    c.offA = (AKR ? c.m0 * 4u : (unsigned)(c.m0 * g.lda * 4)) + c.kt * kadvA;
    c.offB = (BKR ? c.n0 * 4u : (unsigned)(c.n0 * g.ldb * 4)) + c.kt * kadvB;
  };
  auto advance = [&](Cursor& c) {
    c.kt++;
    if (c.kt == nk) { c.kt = 0; c.seq++; place(c); }
    else { c.offA += kadvA; c.offB += kadvB; }
  };
  Cursor ld{0, STREAMK ? it_b % nk : 0u, 0, 0, 0, 0}, cp = ld;
  place(ld); place(cp);

  u32x4 P[NP];
  auto gload_one = [&](int i, const Cursor& c) {
    if (i < NPA) P[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, c.offA + (unsigned)i * ioffA, 0);
    else P[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, c.offB + (unsigned)(i - NPA) * ioffB, 0);
  };
  auto lwrite_one = [&](int i) {
    if (i < NPA) *reinterpret_cast<u32x4*>(ldsA + swA + i * SW_STEP_A) = P[i];
    else *reinterpret_cast<u32x4*>(ldsB + swB + (i - NPA) * SW_STEP_B) = P[i];
  };
  f32x4 fa[4][4], fb[4][4];      // [j][r]: k-contiguous operand: r = 16-row tile, components = 4 k-steps; rows-are-k: r = k-step, components = 4 tiles
  auto fread = [&](int j, int r, bool isB) {
    if (!isB) fa[j][r] = *reinterpret_cast<const f32x4*>(ldsA + fra + (AKR ? (j * 8192 + r * 512) : (r * 256 + j * 64)));
    else fb[j][r] = *reinterpret_cast<const f32x4*>(ldsB + frb + (BKR ? (j * 8192 + r * 512) : (r * 256 + j * 64)));
  };
  f32x4 acc[TMW][4];
  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};   // DB: column sums of A over this segment's k range, rows m0 + 64 wy + 4 c16 + {0..3}, this lane's k-steps

  // ---- prologue: k-tile 0 -> LDS, k-tile 1 -> registers, fragments j = 0 of k-tile 0 ----
#pragma unroll
  for (int i = 0; i < NP; i++) gload_one(i, ld);
  advance(ld);
#pragma unroll
  for (int i = 0; i < NP; i++) lwrite_one(i);
#pragma unroll
  for (int i = 0; i < NP; i++) gload_one(i, ld);
  advance(ld);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  SK_PIN();
#pragma unroll
  for (int r = 0; r < 4; r++) { if (r < TMW) fread(0, r, false); fread(0, r, true); }
  SK_PIN();

  // outer loop: the output tiles (stream-K: segments) of this workgroup; inner loop: their k-tiles.  The operand stream (cursor
  // ld, two k-tiles ahead) does not know about the nest.
  for (unsigned it = 0; it < n_it;) {
    const unsigned seg = (nk - cp.kt) < (n_it - it) ? (nk - cp.kt) : (n_it - it);
#pragma unroll
    for (int i = 0; i < TMW; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 7" ::: "memory");
    for (unsigned kk = 0; kk < seg; kk++) {
      sk_static_for<NM>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        // s -> (k-group j, k-step e, row tile tm, column tile tn): tn fastest, then tm, then e, then j
        constexpr int tn = s & 3, tm = (s >> 2) % TMW, e = (s / (4 * TMW)) & 3, j = s / (16 * TMW);
        const float av = AKR ? fa[j][e][tm] : fa[j][tm][e];
        const float bv = BKR ? fb[j][e][tn] : fb[j][tn][e];
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[tm][tn]) : "v"(av), "v"(bv));
        if constexpr (DB && (s & 15) == 1) {      // one k-step of four row tiles, behind the first MFMA that used it.  Spelled as
          // instructions: left to itself hipcc gathers the 16 adds of a k-tile in one place (fragments parked in AGPRs meanwhile) and
          // packs them into v_pk_add_f32, which costs several times a v_add_f32 beside MFMAs
          asm volatile("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %5, %1\n\tv_add_f32 %2, %6, %2\n\tv_add_f32 %3, %7, %3"
                       : "+v"(bsum.x), "+v"(bsum.y), "+v"(bsum.z), "+v"(bsum.w)
                       : "v"(fa[j][e].x), "v"(fa[j][e].y), "v"(fa[j][e].z), "v"(fa[j][e].w));
        }
        // the schedule in the MFMAs' shadows (128-row tiles: the constants of the header comment; 64-row tiles: the same order, scaled)
        constexpr int RPG = TMW + 4;                        // fragment reads per k-group: TMW of A, 4 of B
        constexpr int NRD = 3 * RPG;                        // ... of k-groups 1..3 of this k-tile: at s = 0 .. NRD - 1
        constexpr int BAR1 = TMW == 4 ? 53 : 30;            // every wave has its fragments: the LDS buffer may be overwritten
        constexpr int PSTEP = TMW == 4 ? 11 : 7;            // one (write the next k-tile's piece, load the one after it) pair every PSTEP MFMAs
        constexpr int BAR2 = NM - 13;                       // the next k-tile is in LDS
        if constexpr (s < NRD) {                   // fragments of k-groups 1..3
          constexpr int jj = 1 + s / RPG, gi = s % RPG;
          // order within a group: A0 B0 A1 B1 .. then the remaining B tiles
          if constexpr (gi < 2 * TMW) fread(jj, gi >> 1, (gi & 1) != 0);
          else fread(jj, gi - TMW, true);
        }
        if constexpr (s == BAR1) {
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (s > BAR1 && s <= BAR1 + 1 + (NP - 1) * PSTEP && (s - BAR1 - 1) % PSTEP == 0) {
          constexpr int i = (s - BAR1 - 1) / PSTEP;
          lwrite_one(i);                           // next k-tile: registers -> LDS
          gload_one(i, ld);                        // the one after it -> the same registers
        }
        if constexpr (s == BAR2) {
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (s > BAR2 && s <= BAR2 + RPG) {       // first k-group of the next k-tile, in the order its MFMAs want them
          constexpr int o = s - BAR2 - 1;          // A0 B0 B1 B2 B3 A1 [A2 A3]
          if constexpr (o == 0) fread(0, 0, false);
          else if constexpr (o <= 4) fread(0, o - 1, true);
          else fread(0, o - 4, false);
        }
        SK_PIN();
      });
      advance(ld);
    }
    it += seg;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA result -> v_accvgpr_read
    bool store_tile = true;
    if constexpr (SPLIT) {
      const bool head = cp.kt == 0;
      if (!(head && seg == nk)) {
        typedef unsigned long long u64;
        const unsigned it0 = (it_b / nk + cp.seq) * nk;                  // the tile's first iteration in the flat space (place())
        const unsigned r0 = it0 / ipw, r1 = (it0 + nk - 1) / ipw;        // the ranges it spans: part p is range r0 + p's share
        const unsigned nparts = r1 - r0 + 1;
        // 1. this part's accumulators -> its slot
        u64* slot = reinterpret_cast<u64*>(g.sk_slots + (size_t)(head ? 2u * rng + 1u : 2u * rng) * (SK_BM * SK_BN));
#pragma unroll
        for (int tm = 0; tm < TMW; tm++)
#pragma unroll
          for (int tn = 0; tn < 4; tn++) {
            const f32x4 v = acc[tm][tn];
            u64* q2 = slot + ((tm * 4 + tn) * 256 + tid) * 2;
            __hip_atomic_store(q2, ((u64)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q2 + 1, ((u64)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // 2. arrive; the value the add returns says whether every other part is in memory already
        unsigned* const bc = reinterpret_cast<unsigned*>(sk_lds + LDS_A + (BKR ? SK_LDS_KR : SK_LDS_KC));      // one word behind the operand images
        if (tid == 0) *bc = __hip_atomic_fetch_add(g.sk_cnt + r0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const bool last = *bc == nparts - 1;
        __syncthreads();                      // (the word is free again)
        if (!last) {
          store_tile = false;
        } else {
          if (tid == 0) __hip_atomic_store(g.sk_cnt + r0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch on this stream
          // 3. the parts in k order.  The head's own part is in its registers; any other last arriver reads every part, its own included
          //    (the bits it stored), so that the order of the adds does not depend on who came last
          for (unsigned p = head ? 1u : 0u; p < nparts; p++) {
            const u64* sl = reinterpret_cast<const u64*>(g.sk_slots + (size_t)(p == 0 ? 2u * r0 + 1u : 2u * (r0 + p)) * (SK_BM * SK_BN));
#pragma unroll
            for (int tm = 0; tm < TMW; tm++)
#pragma unroll
              for (int tn = 0; tn < 4; tn++) {
                const u64* q2 = sl + ((tm * 4 + tn) * 256 + tid) * 2;
                const u64 lo = __hip_atomic_load(q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 hi = __hip_atomic_load(q2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const f32x4 v = f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
                if (p == 0) acc[tm][tn] = v; else acc[tm][tn] += v;
              }
          }
        }
      }
    }
    // ---- epilogue of the tile / segment: lane (c16, q) of wave (wy, wx) holds, for tm, i in 0..3, the four columns
    //      n0 + 64 wx + 4 c16 + {0..3} of row m0 + 64 wy + 16 q + 4 i + tm
    if (store_tile) {
      const int nn = (int)cp.n0 + 64 * wx + 4 * c16;
      const bool relu_fast = g.act == FFH_AC_MODE_RELU, none_fast = g.act == FFH_AC_MODE_NONE;      // uniform
      (void)relu_fast; (void)none_fast;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == SK_EPI_FWD) {
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.bias), 0, g.bias_bytes, 0x00020000);
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsBias, (unsigned)nn * 4u, 0, 0);     // absent bias: extent 0, the load returns 0
        bv = __builtin_bit_cast(f32x4, t);
      }
      // dX: the relu' mask (x of this layer) and, in the accumulating form, C's previous content are fetched for all 16 row groups
      // BEFORE the first store: left inside the loop below each load sat behind the store in front of it (the compiler cannot tell
      // the two apart, and s_waitcnt vmcnt(0) counts stores too) -- sixteen dependent round trips per tile on a SIMD whose only wave
      // this is
      // DX_CMAP: where this lane's four columns go.  Destinations are runs of whole tensors (a table's slot in the send buffer, the
      // bottom MLP's gradient), so four consecutive columns normally share one and the store stays 16 bytes wide; where a group
      // straddles two destinations (or one is unaligned) its four columns are stored one by one
      ffh_col_dest cd0{nullptr, 0}, cd1{nullptr, 0}, cd2{nullptr, 0}, cd3{nullptr, 0};
      bool cvec = false;
      if constexpr (EPI == SK_EPI_DX_CMAP) {
        cd0 = g.colmap[nn]; cd3 = g.colmap[nn + 3];
        cvec = cd3.base == cd0.base + 3 && cd3.ld == cd0.ld && ((((uintptr_t)cd0.base) | (uintptr_t)(cd0.ld * 4)) & 15) == 0;
        if (!cvec) { cd1 = g.colmap[nn + 1]; cd2 = g.colmap[nn + 2]; }
      }
      f32x4 mk[TMW][4], cold[TMW][4];
      f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};     // DX_STORE with g.colsum: this lane's four columns summed over its 16 rows
      (void)csum;
      if constexpr (EPI == SK_EPI_DX_STORE || EPI == SK_EPI_DX_ADD) {
#pragma unroll
        for (int tm = 0; tm < TMW; tm++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int mm = (int)cp.m0 + 16 * TMW * wy + TMW * (4 * q + i) + tm;
            if (g.mask_bytes) mk[tm][i] = *reinterpret_cast<const f32x4*>(g.mask + (int64_t)mm * g.ldmask + nn);      // uniform
            if constexpr (EPI == SK_EPI_DX_ADD) cold[tm][i] = *reinterpret_cast<const f32x4*>(g.C + (int64_t)mm * g.ldc + nn);
          }
      }
#pragma unroll
      for (int tm = 0; tm < TMW; tm++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int mm = (int)cp.m0 + 16 * TMW * wy + TMW * (4 * q + i) + tm;
          f32x4 v = f32x4{acc[tm][0][i], acc[tm][1][i], acc[tm][2][i], acc[tm][3][i]};
          float* cptr = g.C + (int64_t)mm * g.ldc + nn;
          if constexpr (EPI == SK_EPI_DW_ATOMIC) {
            // through a per-wave LDS image so that one atomic instruction covers 256 contiguous bytes of one row (the shape at
            // which memory-side float atomics run at their full rate; 4-byte pieces 16 bytes apart run ~10x slower)
            (void)cptr; (void)v;          // below: 16 rows at a time
          } else if constexpr (EPI == SK_EPI_FWD) {
            v += bv;
            // ReLU / none without act_apply's chain of (uniform) branches per element: 64 elements x 3 branches per lane and tile were
            // ~2 us of every tile with nothing else to run on the SIMD; same values (v > 0 ? v : 0)
            if (relu_fast) { v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f; }
            else if (!none_fast) { v.x = act_apply(v.x, g.act); v.y = act_apply(v.y, g.act); v.z = act_apply(v.z, g.act); v.w = act_apply(v.w, g.act); }
            *reinterpret_cast<f32x4*>(cptr) = v;
          } else if constexpr (EPI == SK_EPI_DX_CMAP) {
            if (cvec) {
              *reinterpret_cast<f32x4*>(cd0.base + (int64_t)mm * cd0.ld) = v;
            } else {
              cd0.base[(int64_t)mm * cd0.ld] = v.x; cd1.base[(int64_t)mm * cd1.ld] = v.y;
              cd2.base[(int64_t)mm * cd2.ld] = v.z; cd3.base[(int64_t)mm * cd3.ld] = v.w;
            }
          } else {
            if (g.mask_bytes) {        // uniform
              const f32x4 m4 = mk[tm][i];
              v.x = m4.x > 0.0f ? v.x : 0.0f; v.y = m4.y > 0.0f ? v.y : 0.0f; v.z = m4.z > 0.0f ? v.z : 0.0f; v.w = m4.w > 0.0f ? v.w : 0.0f;
            }
            if constexpr (EPI == SK_EPI_DX_ADD) v += cold[tm][i];
            if constexpr (EPI == SK_EPI_DX_STORE) csum += v;
            *reinterpret_cast<f32x4*>(cptr) = v;
          }
        }
      if constexpr (EPI == SK_EPI_DX_STORE) {
        if (g.colsum) {            // uniform: the lane's 16 rows are summed above; the four lane groups hold rows 16 q ..: fold them, one atomic per column and wave
          f32x4 t = csum;
#pragma unroll
          for (int cc = 0; cc < 4; cc++) { t[cc] += __shfl_xor(t[cc], 16); t[cc] += __shfl_xor(t[cc], 32); }
          if (q == 0) { float* dp = g.colsum + nn; atomicAdd(dp + 0, t.x); atomicAdd(dp + 1, t.y); atomicAdd(dp + 2, t.z); atomicAdd(dp + 3, t.w); }
        }
      }
    }
    if constexpr (EPI == SK_EPI_DW_ATOMIC) {
      // through a per-wave LDS image so that one atomic instruction covers 256 contiguous bytes of one row
      float* const ep = reinterpret_cast<float*>(sk_lds + LDS_A + (BKR ? SK_LDS_KR : SK_LDS_KC) + wave * SK_EP_WAVE);
      float* crow = g.C + (int64_t)((int)cp.m0 + 64 * wy) * g.ldc + (int)cp.n0 + 64 * wx + lane;
#pragma unroll
      for (int p = 0; p < 4; p++) {            // rows 16 p .. 16 p + 15 of the wave's result are held by the lanes with q == p
        if (q == p) {
#pragma unroll
          for (int tm = 0; tm < 4; tm++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              *reinterpret_cast<f32x4*>(ep + (4 * i + tm) * SK_EP_LD + 4 * c16) = f32x4{acc[tm][0][i], acc[tm][1][i], acc[tm][2][i], acc[tm][3][i]};
        }
        __builtin_amdgcn_wave_barrier();       // the wave's own image: no workgroup barrier (a wave's ds operations complete in order)
#pragma unroll
        for (int r = 0; r < 16; r++) atomicAdd(crow + (int64_t)(16 * p + r) * g.ldc, ep[r * SK_EP_LD + lane]);
        __builtin_amdgcn_wave_barrier();
      }
      if constexpr (DB) {
        if (cp.n0 == 0 && wx == 0) {          // wave-uniform: the first column of tiles owns the bias gradient
          f32x4 t = bsum;
#pragma unroll
          for (int c = 0; c < 4; c++) { t[c] += __shfl_xor(t[c], 16); t[c] += __shfl_xor(t[c], 32); }
          if (q == 0) {
            float* dbp = g.db + (int)cp.m0 + 64 * wy + 4 * c16;
            atomicAdd(dbp + 0, t.x); atomicAdd(dbp + 1, t.y); atomicAdd(dbp + 2, t.z); atomicAdd(dbp + 3, t.w);
          }
        }
        bsum = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    cp.kt += seg;
    if (cp.kt == nk) { cp.kt = 0; cp.seq++; place(cp); }
  }
}

template <typename K>
bool sk_set_lds(K kern, int bytes) {
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

inline bool sk_aligned(const void* p, int64_t ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

}  // namespace

namespace ffh_gemm {

namespace {
struct SkPlan { int64_t lda, ldb, a_bytes, b_bytes; int G; bool split; int tmw; };
bool sk_plan(const ffh_ctx* c, const GemmArgs& g, int form, SkPlan& p) {
  static const int off = FFH_LAB_INT("FFH_GEMM_NO_SK", 0);   // A/B switch (tools/ab.sh)
  if (off) return false;
  {
    // The backward GEMMs of a layer with fewer than 256 x 256 weights stay with the register-staged kernels: their workgroups
    // (33-37 KB of LDS, 100-160 registers) fit on a CU beside a persistent one, a second persistent workgroup does not (registers).  In
    // the DLRM step that layer is the bottom MLP's last one, whose backward is launched while the first top layer's stream-K weight
    // gradient holds a workgroup on every CU: as a persistent kernel it waited ~1.7 ms for CUs and the rest of the bottom MLP's
    // backward behind it (Terabyte step 8.07-8.10 -> 8.01-8.03 ms; at 140 K the two 512 x 256 layers move too: 8.13-8.15).
    // FFH_SK_MIN_WEIGHTS / _BWD: A/B switches (all forms / the two backward forms).
    static const int64_t min_w = FFH_LAB_I64("FFH_SK_MIN_WEIGHTS", 0);
    static const int64_t min_wb = FFH_LAB_I64("FFH_SK_MIN_WEIGHTS_BWD", 65536);
    const int64_t nw = form == SK_FORM_DW ? (int64_t)g.M * g.N : (int64_t)g.N * g.K;
    if (nw < min_w || (form != SK_FORM_FWD && nw < min_wb)) return false;
  }
  if (g.M % SK_BM || g.N % SK_BN || g.K % SK_BK || g.M <= 0 || g.N <= 0 || g.K <= 0) return false;      // (64-row tiles: M a multiple of 128 as well -- every other layer has it)
  const bool akr = form == SK_FORM_DW, bkr = form != SK_FORM_FWD;
  p.lda = akr ? g.sAk : g.sAm; p.ldb = bkr ? g.sBk : g.sBn;
  if ((akr ? g.sAm : g.sAk) != 1 || (bkr ? g.sBn : g.sBk) != 1) return false;
  if (!sk_aligned(g.A, p.lda) || !sk_aligned(g.B, p.ldb) || (!g.colmap && !sk_aligned(g.C, g.ldc))) return false;
  if (g.mask && !sk_aligned(g.mask, g.ldmask)) return false;
  if (g.bias && ((uintptr_t)g.bias & 15)) return false;
  if (g.act_y || g.fuse || g.splitk > 1) return false;  // masking-while-loading forms stay with linear.hip
  if (g.colmap && (form != SK_FORM_DX || g.epi != EPI_STORE || g.mask)) return false;     // the column map: plain stored data gradients only
  p.a_bytes = ((akr ? (int64_t)(g.K - 1) : (int64_t)(g.M - 1)) * p.lda + (akr ? g.M : g.K)) * 4;
  p.b_bytes = ((bkr ? (int64_t)(g.K - 1) : (int64_t)(g.N - 1)) * p.ldb + (bkr ? g.N : g.K)) * 4;
  if (p.a_bytes >= (1LL << 32) || p.b_bytes >= (1LL << 32)) return false;  // 32-bit buffer offsets
  p.G = c->num_cus & ~7;
  if (form == SK_FORM_DW && c->dw_cu_reserve > 0 && p.G - c->dw_cu_reserve >= 64) p.G -= c->dw_cu_reserve;     // ffh_ctx_set_dw_cu_reserve
  if (p.G < 8) return false;
  const int64_t ntiles = (int64_t)(g.M / SK_BM) * (g.N / SK_BN), nk = g.K / SK_BK;
  if (ntiles * nk >= (1LL << 31)) return false;
  p.tmw = 4;
  if (form == SK_FORM_DW) {
    if (c->deterministic || g.epi != EPI_ATOMIC) return false;     // its partial tiles meet by atomics
    static const int dw_min_it = FFH_LAB_INT("FFH_SK_DW_MIN_IT", 8);       // A/B switch
    if (ntiles * nk < (int64_t)dw_min_it * p.G) return false;      // at least eight k-tiles per workgroup (with four the split-K kernels of linear.hip win:
                                                                   // 8192 x 512 -> 256: 32.3 vs 37.0 us; with eight this one does: 4096 x 1024 -> 512: 48.7 vs 52.5)
  } else {
    if (g.epi != EPI_STORE && g.epi != EPI_ADD) return false;
    // whole tiles when they fill whole rounds of workgroups (or nearly: a split costs a partial tile's round trip through memory);
    // otherwise stream-K with the fix-up of the kernel's SPLIT form when every workgroup still gets >= 8 k-tiles
    static const int no_split = FFH_LAB_INT("FFH_SK_NO_SPLIT", 0);     // A/B switch (tools/ab.sh)
    const int64_t rounds = (ntiles + p.G - 1) / p.G;
    const int64_t idle_it = (rounds * p.G - ntiles) * nk / p.G;      // k-tile iterations per workgroup the last round wastes
    // less than one round of 128-row tiles, one (nearly) full round of 64-row tiles: those, whole, with the full reduction each
    // (1024 -> 512 forward at 4096 samples: 128 tiles / 256)
    static const int no_t64 = FFH_LAB_INT("FFH_SK_NO_T64", 0);        // A/B switch
    p.tmw = 4;
    if (!no_t64 && ntiles < p.G && g.M % 64 == 0 && (g.epi == EPI_STORE || g.epi == EPI_ADD) && nk >= 8) {
      const int64_t nt64 = (int64_t)(g.M / 64) * (g.N / SK_BN);
      if (nt64 <= p.G && nt64 * 100 >= (int64_t)p.G * 80) { p.tmw = 2; p.split = false; return true; }
    }
    p.split = !no_split && idle_it >= 2 && ntiles * nk >= 8LL * p.G && g.epi == EPI_STORE;    // (four k-tiles per workgroup: 8192 x 512 -> 256 forward 33.3 us split, 27.8 on the LDS-DMA kernel)
    if (!p.split) {
      if (ntiles < p.G) return false;
      if (ntiles * 100 < rounds * p.G * 80) return false;          // whole tiles only: the last round must be nearly full
    }
  }
  return true;
}

// the partial-tile slots of the SPLIT form for launches on stream s: ctx-owned scratch reserved by ffh_ctx_reserve_scratch(ctx, s)
// (runtime.hip) -- a compute entry point never allocates.  No set for this stream: the form is not offered.
bool sk_slots_for(ffh_ctx* c, hipStream_t s, float** slots, unsigned** cnt) {
  for (int i = 0; i < c->nscratch; i++)
    if (c->scratch[i].stream == (void*)s && c->scratch[i].sk_slots) { *slots = c->scratch[i].sk_slots; *cnt = c->scratch[i].sk_cnt; return true; }
  return false;
}
}  // namespace

bool gemm_sk_serves(const ffh_ctx* c, const GemmArgs& g, int form) { SkPlan p; return sk_plan(c, g, form, p); }

// 1: launched; 0: not this kernel's shape (nothing launched); < 0: error
int launch_gemm_sk(ffh_ctx* c, const GemmArgs& g, int form, ffh_stream s, const char* name) {
  SkPlan p{};
  if (!sk_plan(c, g, form, p)) return 0;
  float* slots = nullptr; unsigned* flags = nullptr;
  if (p.split && !sk_slots_for(c, as_stream(s), &slots, &flags)) {
    // no scratch reserved for this stream (ffh_ctx_reserve_scratch): the whole-tile form where it serves, else not this kernel's launch --
    // said in the route, so that a caller who launches on a stream of its own sees why a slower form ran
    p.split = false;
    { char tok[96]; snprintf(tok, sizeof tok, "%s|no_scratch_on_this_stream", name); ffh_route_add(c, tok); }
    const int64_t ntiles = (int64_t)(g.M / SK_BM) * (g.N / SK_BN), rounds = (ntiles + p.G - 1) / p.G;
    if (ntiles < p.G || ntiles * 100 < rounds * p.G * 80) return 0;
  }
  const int64_t lda = p.lda, ldb = p.ldb, a_bytes = p.a_bytes, b_bytes = p.b_bytes;
  const int G = p.G;
  SkArgs a{};
  a.A = g.A; a.B = g.B; a.C = g.C; a.bias = g.bias; a.mask = g.mask; a.db = g.db; a.colmap = g.colmap; a.colsum = (form == SK_FORM_DX && g.epi == EPI_STORE && !g.colmap) ? g.colsum : nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = g.ldc; a.ldmask = g.ldmask;
  a.M = g.M; a.N = g.N; a.K = g.K; a.act = g.act;
  a.a_bytes = (unsigned)a_bytes; a.b_bytes = (unsigned)b_bytes;
  a.bias_bytes = g.bias ? (unsigned)g.N * 4u : 0u;
  a.mask_bytes = g.mask ? 1u : 0u;
  if (p.split) { a.sk_slots = slots; a.sk_cnt = flags; }
  // The kernels need more than 64 KB of dynamic LDS; hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of the
  // function, and one process may hold a ctx per device (ffh_ctx_default): set once per (kernel, device).  A launch that still
  // fails (attribute refused, no such resources) is not an error of the call: 0 = "not served", linear.hip's kernels take the layer.
#define FFH_SK_LAUNCH(AKR, BKR, EPI, LDSB, ...)                                                                  \
  {                                                                                                              \
    auto kern = gemm_sk_kernel<AKR, BKR, EPI, ##__VA_ARGS__>;                                                    \
    static std::atomic<signed char> ok[64];      /* 0: not tried on this device, 1: set, -1: refused */          \
    const int dev = c->device & 63;                                                                              \
    if (ok[dev].load(std::memory_order_acquire) == 0) ok[dev].store(sk_set_lds(kern, LDSB) ? 1 : -1, std::memory_order_release); \
    if (ok[dev].load(std::memory_order_acquire) < 0) return 0;                                                   \
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(256), LDSB, as_stream(s), a);                               \
  }
  constexpr int BCW = 64;      // SPLIT: the arrival broadcast word behind the operand images
  constexpr int KC64 = sk_lds_kc_a(2);
  if (form == SK_FORM_FWD && p.tmw == 2) FFH_SK_LAUNCH(false, false, SK_EPI_FWD, KC64 + SK_LDS_KC, false, false, 2)
  else if (form == SK_FORM_DX && p.tmw == 2 && g.colmap) FFH_SK_LAUNCH(false, true, SK_EPI_DX_CMAP, KC64 + SK_LDS_KR, false, false, 2)
  else if (form == SK_FORM_DX && p.tmw == 2 && g.epi == EPI_STORE) FFH_SK_LAUNCH(false, true, SK_EPI_DX_STORE, KC64 + SK_LDS_KR, false, false, 2)
  else if (form == SK_FORM_DX && p.tmw == 2) FFH_SK_LAUNCH(false, true, SK_EPI_DX_ADD, KC64 + SK_LDS_KR, false, false, 2)
  else if (form == SK_FORM_FWD && p.split) FFH_SK_LAUNCH(false, false, SK_EPI_FWD, 2 * SK_LDS_KC + BCW, false, true)
  else if (form == SK_FORM_DX && p.split && g.colmap) FFH_SK_LAUNCH(false, true, SK_EPI_DX_CMAP, SK_LDS_KC + SK_LDS_KR + BCW, false, true)
  else if (form == SK_FORM_DX && p.split) FFH_SK_LAUNCH(false, true, SK_EPI_DX_STORE, SK_LDS_KC + SK_LDS_KR + BCW, false, true)
  else if (form == SK_FORM_FWD) FFH_SK_LAUNCH(false, false, SK_EPI_FWD, 2 * SK_LDS_KC)
  else if (form == SK_FORM_DW && g.db) FFH_SK_LAUNCH(true, true, SK_EPI_DW_ATOMIC, 2 * SK_LDS_KR + 4 * SK_EP_WAVE, true)
  else if (form == SK_FORM_DW) FFH_SK_LAUNCH(true, true, SK_EPI_DW_ATOMIC, 2 * SK_LDS_KR + 4 * SK_EP_WAVE)
  else if (g.colmap) FFH_SK_LAUNCH(false, true, SK_EPI_DX_CMAP, SK_LDS_KC + SK_LDS_KR)
  else if (g.epi == EPI_STORE) FFH_SK_LAUNCH(false, true, SK_EPI_DX_STORE, SK_LDS_KC + SK_LDS_KR)
  else FFH_SK_LAUNCH(false, true, SK_EPI_DX_ADD, SK_LDS_KC + SK_LDS_KR)
#undef FFH_SK_LAUNCH
  {
    // a launch the device refuses for its CONFIGURATION enqueued nothing: "not served", the caller falls through to linear.hip.  Anything
    // else (a sticky fault of earlier work surfacing here) is the caller's to see
    const hipError_t e = hipGetLastError();
    if (e == hipErrorInvalidValue || e == hipErrorInvalidConfiguration || e == hipErrorLaunchOutOfResources || e == hipErrorSharedObjectInitFailed) return 0;
    if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  }
  { char tok[96]; snprintf(tok, sizeof tok, "%s|sk_%dx128x64%s%s%s|wgs=%d", name, 32 * p.tmw, g.colmap ? "|colmap" : "", p.split ? "|streamk" : "", a.colsum ? "|colsum" : "", G); ffh_route_add(c, tok); }
  return 1;
}

}  // namespace ffh_gemm
