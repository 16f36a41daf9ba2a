// mlp_chain.hip -- a chain of NARROW Linear layers (every width <= 512: DLRM's bottom MLP 13-512-256-128, the Kaggle shape's
// 13-512-256-64-16 and 432-512-256-1) as three launches instead of three per layer:
//   mlp_chain_fwd_kernel   all layers' forward for a block of 16 / 32 samples: the activations between the layers stay in LDS
//                          (and are written out once, coalesced, because the backward reads them), the weights stream from L2
//                          straight into MFMA operand registers -- each weight is used by exactly one wave of a workgroup, so LDS
//                          would only add a copy;
//   mlp_chain_dx_kernel    the data-gradient chain top -> bottom for the same block of samples (dy_l -> dy_(l-1) through LDS, the
//                          relu' of the layer below applied where the gradient is produced);
//   mlp_chain_dw_kernel    every layer's weight / bias gradient in ONE launch: 64 x 64 blocks of dW x batch splits, eight waves of
//                          a workgroup share a block, meet in LDS and add the block to dW with row-contiguous atomics.
// Why: at 2048-8192 samples per GPU these layers are 5-30 us kernels whose launches depend on each other; in the per-rank step of
// the 8-GPU job (4096 samples) the bottom MLP's backward + forward were ~12 launches and ~100 us of a 1.18 ms step with the matrix
// pipe idle (profiles/r04_terabyte_b4096_plain_step_timeline.txt, +709 .. +841 us), 18 us of arithmetic at the fp32 MFMA peak.
// Arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32), one fmaf chain per output element with k visited in the order 16j + 4q + e
// (forward, dX) / 4s + q (dW) -- a fixed order per kernel, different from the per-layer kernels'; parity tests hold it to 1e-5 of
// the term mass against the oracle, like every other GEMM here.
//
// Replaces, for such a chain, the per-layer Linear::forward_kernel / backward_kernel calls [ref: src/ops/linear.cu:425-465,610-660];
// the reference's own precedent for several operators in one task is FusedOp [ref: src/ops/fused.cu:283-400].
#include "linear_gemm.h"

using namespace ffh_gemm;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_MAXL = FFH_CHAIN_MAX_LAYERS;
constexpr int CH_MAXW = FFH_CHAIN_MAX_WIDTH;
constexpr int CH_THREADS = 512;               // 8 waves: two per SIMD, one covers the other's operand latency
constexpr unsigned CH_OOB = 0x80000000u;      // a buffer offset past every descriptor's extent: the load returns 0

// LDS row stride (floats) of an activation block of width w: the width rounded up to whole 16-deep k-groups + 8.  An odd
// multiple of 8 floats puts the sixteen rows one ds_read_b128 fragment read touches (lane (c, q): row c, 16 bytes at k0 + 4q) on
// sixteen different 4-bank groups: conflict-free, as linear_sk.hip's k-contiguous image.
__host__ __device__ constexpr int ch_stride(int w) { return ((w + 15) & ~15) + 8; }

struct ChLayer {
  const float* w; const float* bias;
  float* y; float* dy; float* dw; float* db;
  int64_t ldy, lddy;
  int ldw, in, out, act;
  unsigned w_bytes;        // extent of w for the buffer descriptor
  int w_vec;               // rows of w are 16-byte aligned and in % 4 == 0: one dwordx4 load per fragment
};

struct ChainFwdArgs {
  const float* x; int64_t ldx; int64_t batch;
  int n; int buf1_off;     // floats: start of the second activation buffer
  int stage_off;           // floats: start of the waves' weight images (behind the two activation buffers)
  int x_vec;
  ChLayer L[CH_MAXL];
};

struct ChainDxArgs {
  float* dx; int64_t lddx;             // the chain input's gradient, or null
  const float* xmask; int64_t ldxmask; // FFH_LINEAR_DX_MASK_BY_X: the chain input (a ReLU output), or null
  int64_t batch;
  int n; int buf1_off;
  int dx_add;                          // dx += (no FFH_LINEAR_DX_OVERWRITE)
  int top_live;                        // the top layer's activation derivative is still to be applied to its dy (in place)
  ChLayer L[CH_MAXL];
};

struct DwLayer {
  const float* dy; const float* x; float* dw; float* db;
  int64_t lddy, ldx;
  int ldw, M, N;           // dW is [M = out][N = in]
  int nbn;                 // 64-wide column blocks
  int first_item;          // index of this layer's first (row block, column block) item
  int a_vec, b_vec;
};
struct ChainDwArgs {
  int64_t batch;
  int n, nitems, S;
  int64_t rows_per_split;  // multiple of 32
  float* slots;            // deterministic mode: workgroup (item, split) leaves its 64 x 64 block (+ 64 bias sums) in slot item * S + split of
                           // kSkinnyWsRow floats instead of adding it to dW; mlp_chain_dw_reduce_kernel adds the splits in order.  Null: atomics
  DwLayer L[CH_MAXL];
};

#define CH_PIN() __builtin_amdgcn_sched_barrier(0)
#define CH_SGPR(x) asm volatile("" : "+s"(x))

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 bload4(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ float bload1(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward: one layer for the workgroup's R rows.  A = the input block in LDS ([R][Sin], zero beyond K up to the k-group), B = w
// [N][ldw] from memory: lane (c, q) of the wave that owns output tile t holds w[16 t + c][16 j + 4q .. + 3] for k-group j -- the
// four k-steps of one 16-wide tile (legal because the A fragment uses the same k permutation).  Ring of D k-groups in flight.
template <int RT, int NT, bool VEC>
__device__ __forceinline__ void ch_fwd_layer(const ChLayer& L, const float* in, float* out, const int wave, const int c16, const int q) {
  const int K = L.in, N = L.out;
  const int Sin = ch_stride(K), Sout = ch_stride(N);
  const int ntiles = (N + 15) >> 4, tpw = (ntiles + 7) >> 3, kgs = (K + 15) >> 4;
  const int t0 = wave * tpw;
  constexpr int D = NT >= 4 ? 4 : 8;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w), 0, L.w_bytes, 0x00020000);
  unsigned rowoff[NT];      // byte offset of this lane's weight row per tile, or CH_OOB
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int n = (t0 + t) * 16 + c16;
    rowoff[t] = (t < tpw && n < N) ? (unsigned)(n * L.ldw) * 4u : CH_OOB;
  }
  auto loadB = [&](int kg, int t) -> f32x4 {
    const int k = kg * 16 + 4 * q;
    if constexpr (VEC) {
      const unsigned off = (rowoff[t] != CH_OOB && k < K) ? rowoff[t] + (unsigned)k * 4u : CH_OOB;
      return bload4(rs, off);
    }
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const unsigned off = (rowoff[t] != CH_OOB && k + e < K) ? rowoff[t] + (unsigned)(k + e) * 4u : CH_OOB;
      v[e] = bload1(rs, off);
    }
    return v;
  };
  f32x4 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int t = 0; t < NT; t++) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 Bf[D][NT];
#pragma unroll
  for (int d = 0; d < D; d++)
#pragma unroll
    for (int t = 0; t < NT; t++) Bf[d][t] = loadB(d, t);
  const float* arow = in + c16 * Sin + 4 * q;
  auto step = [&](int kg, int d, bool refill) {
    f32x4 A[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) A[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 16 * Sin + kg * 16);
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt][t] = mfma4(A[rt][e], Bf[d][t][e], acc[rt][t]);
    if (refill) {
#pragma unroll
      for (int t = 0; t < NT; t++) Bf[d][t] = loadB(kg + D, t);
    }
    CH_PIN();       // the refill stays HERE, D - 1 steps of MFMAs ahead of its use (left alone the scheduler sinks a block's loads to its end)
  };
  int kg0 = 0;
  for (; kg0 + D <= kgs; kg0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) step(kg0 + d, d, true);
  }
#pragma unroll
  for (int d = 0; d < D; d++)
    if (kg0 + d < kgs) step(kg0 + d, d, false);     // uniform
  // epilogue: lane (c, q) holds rows 4q + i of column 16 t + c; bias + activation, into the next layer's input block (pad columns 0)
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (t < tpw && t0 + t < ntiles) {     // uniform
      const int col = (t0 + t) * 16 + c16;
      const float b = (L.bias && col < N) ? L.bias[col] : 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float v = act_apply(acc[rt][t][i] + b, L.act);
          if (col >= N) v = 0.0f;
          out[(rt * 16 + 4 * q + i) * Sout + col] = v;
        }
    }
  }
}

// forward, weights with 16-byte rows: the same arithmetic with w passing through a wave-private LDS image.  Straight from memory a
// B fragment is 16 bytes from each of SIXTEEN rows of w per 16 lanes -- 64 separate requests per load instruction, which the
// address path serves at a fraction of the rate of a contiguous one (the 512 -> 256 layer alone took 21 us at 4096 samples, 3.4 us of
// MFMAs per wave).  Here a step = one 16-wide tile x one 64-deep k-block: four dwordx4 loads whose 16 lanes read 256 contiguous
// bytes of one row, four ds_write_b128 into the wave's 16 x 64 image (row stride 72 floats: the fragment reads below touch sixteen
// different bank groups), four ds_read_b128 = the fragments of the block's four k-groups.  The image is the wave's own: its DS
// instructions execute in order, so the next step's writes cannot pass this step's reads -- one image, no barrier.  Ring of four
// steps of loads in flight; the next step's image + fragments are issued in front of this step's MFMAs.
constexpr int CH_STAGE_LD = 72;                         // floats per image row
constexpr int CH_STAGE_FLOATS = 16 * CH_STAGE_LD;       // per wave
template <int RT, int NT>
__device__ __forceinline__ void ch_fwd_layer_staged(const ChLayer& L, const float* in, float* out, float* stage, const int wave, const int lane,
                                                    const int c16, const int q) {
  int K = L.in, N = L.out, ldw = L.ldw;
  CH_SGPR(K); CH_SGPR(N); CH_SGPR(ldw);                  // in scalar registers NOW: left to itself the compiler re-reads them from the kernel
                                                         // arguments inside the loop, behind an s_waitcnt lgkmcnt(0) that also drains the LDS queue
  const int Sin = ch_stride(K), Sout = ch_stride(N);
  const int ntiles = (N + 15) >> 4, tpw = (ntiles + 7) >> 3, kgs = (K + 15) >> 4, kbs = (K + 63) >> 6;
  const int t0 = wave * tpw;
  const int steps = kbs * NT;                            // step s: k-block s / NT, tile s % NT
  constexpr int G = RT == 1 ? 4 : 2;                     // steps of loads in flight (registers: 16 per step)
  constexpr int U = 4;                                   // unroll: a multiple of G, NT and 2, so that every register index is a constant
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w), 0, L.w_bytes, 0x00020000);
  const int lr = lane >> 4, lc = (lane & 15) * 4;        // load role: row 4i + lr of the tile, columns lc .. lc + 3 of the k-block
  float* const wr = stage + lr * CH_STAGE_LD + lc;       // where piece i goes: + 4 i rows
  const float* const rd = stage + c16 * CH_STAGE_LD + 4 * q;   // fragment of k-group j of the block: + 16 j
  const float* const arow = in + c16 * Sin + 4 * q;
  auto gload = [&](int s, int t, f32x4 (&P)[4]) {       // t = s % NT, passed as a constant
    const int kb = s / NT, k = kb * 64 + lc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int n = (t0 + t) * 16 + 4 * i + lr;
      const bool ok = s < steps && t < tpw && n < N && k < K;
      P[i] = bload4(rs, ok ? (unsigned)(n * ldw + k) * 4u : CH_OOB);
    }
  };
  f32x4 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int t = 0; t < NT; t++) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 P[G][4], F[2][4], A[RT][4];
#pragma unroll
  for (int g = 0; g < G; g++) gload(g, g % NT, P[g]);
  auto stage_in = [&](f32x4 (&Pg)[4], f32x4 (&Fg)[4]) {      // registers -> image -> fragments
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4*>(wr + 4 * i * CH_STAGE_LD) = Pg[i];
#pragma unroll
    for (int j = 0; j < 4; j++) Fg[j] = *reinterpret_cast<const f32x4*>(rd + 16 * j);
  };
  stage_in(P[0], F[0]);
  gload(G, G % NT, P[0]);
  CH_PIN();
  for (int s0 = 0; s0 < steps; s0 += U) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int s = s0 + u;
      const int t = u % NT;                              // (U is a multiple of NT)
      const int kb = s / NT;
      if (u % NT == 0) {                                 // a new k-block: its four A fragments per row tile
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (kb * 4 + j < kgs) A[rt][j] = *reinterpret_cast<const f32x4*>(arow + rt * 16 * Sin + (kb * 4 + j) * 16);     // uniform
      }
      // the next step's image and fragments, and the refill of its registers, in front of this step's MFMAs
      stage_in(P[(u + 1) % G], F[(u + 1) & 1]);
      gload(s + 1 + G, (u + 1 + G) % NT, P[(u + 1) % G]);
      CH_PIN();
      if (s < steps) {                                   // uniform
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (kb * 4 + j < kgs) {                        // uniform (the last block of a depth that is not a multiple of 64)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
              for (int rt = 0; rt < RT; rt++) acc[rt][t] = mfma4(A[rt][j][e], F[u & 1][j][e], acc[rt][t]);
          }
      }
      CH_PIN();
    }
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (t < tpw && t0 + t < ntiles) {     // uniform
      const int col = (t0 + t) * 16 + c16;
      const float b = (L.bias && col < N) ? L.bias[col] : 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float v = act_apply(acc[rt][t][i] + b, L.act);
          if (col >= N) v = 0.0f;
          out[(rt * 16 + 4 * q + i) * Sout + col] = v;
        }
    }
  }
}

template <int RT>
__global__ __launch_bounds__(CH_THREADS) void mlp_chain_fwd_kernel(const ChainFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  ffh_kernel_prio();
  constexpr int R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  float* const buf0 = ch_lds;
  float* const buf1 = ch_lds + a.buf1_off;
  {     // the block's input rows -> LDS, zero beyond the width (and beyond the batch)
    const int K = a.L[0].in, S = ch_stride(K), Kp = (K + 15) & ~15;
    if (a.x_vec) {
      const int k4 = Kp >> 2;
      for (int idx = tid; idx < R * k4; idx += CH_THREADS) {
        const int r = idx / k4, c = (idx - r * k4) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 + r < a.batch && c < K) v = *reinterpret_cast<const f32x4*>(a.x + (row0 + r) * a.ldx + c);
        *reinterpret_cast<f32x4*>(buf0 + r * S + c) = v;
      }
    } else {
      for (int idx = tid; idx < R * Kp; idx += CH_THREADS) {
        const int r = idx / Kp, c = idx - r * Kp;
        buf0[r * S + c] = (row0 + r < a.batch && c < K) ? a.x[(row0 + r) * a.ldx + c] : 0.0f;
      }
    }
  }
  __syncthreads();
  for (int l = 0; l < a.n; l++) {
    const ChLayer& L = a.L[l];
    const float* in = (l & 1) ? buf1 : buf0;
    float* out = (l & 1) ? buf0 : buf1;
    const int tpw = ((((L.out + 15) >> 4) + 7) >> 3);
    if (L.w_vec) {      // (uniform; the ragged form -- rows of w not 16-byte aligned, DLRM's 13-wide first layer -- loads dwords)
      float* const stage = ch_lds + a.stage_off + wave * CH_STAGE_FLOATS;
      if (tpw <= 1) ch_fwd_layer_staged<RT, 1>(L, in, out, stage, wave, lane, c16, q);
      else if (tpw <= 2) ch_fwd_layer_staged<RT, 2>(L, in, out, stage, wave, lane, c16, q);
      else ch_fwd_layer_staged<RT, 4>(L, in, out, stage, wave, lane, c16, q);
    } else {
      if (tpw <= 1) ch_fwd_layer<RT, 1, false>(L, in, out, wave, c16, q);
      else if (tpw <= 2) ch_fwd_layer<RT, 2, false>(L, in, out, wave, c16, q);
      else ch_fwd_layer<RT, 4, false>(L, in, out, wave, c16, q);
    }
    __syncthreads();
    // the layer's output, once, coalesced (the backward reads it: relu' and the weight gradient's operand)
    const int N = L.out, S = ch_stride(N);
    if ((N & 3) == 0 && (L.ldy & 3) == 0 && (((uintptr_t)L.y) & 15) == 0) {
      const int n4 = N >> 2;
      for (int idx = tid; idx < R * n4; idx += CH_THREADS) {
        const int r = idx / n4, c = (idx - r * n4) * 4;
        if (row0 + r < a.batch) *reinterpret_cast<f32x4*>(L.y + (row0 + r) * L.ldy + c) = *reinterpret_cast<const f32x4*>(out + r * S + c);
      }
    } else {
      for (int idx = tid; idx < R * N; idx += CH_THREADS) {
        const int r = idx / N, c = idx - r * N;
        if (row0 + r < a.batch) L.y[(row0 + r) * L.ldy + c] = out[r * S + c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// data gradient of one layer for the workgroup's R rows: dX = dy w, dy [R][K = out] in LDS, w [K][N = in] from memory as it lies:
// lane (c, q) loads w[16 j + 4q + e][64 g + 4c .. + 3] -- one k-step of FOUR 16-wide tiles (tile t holds columns 4c + t), so a lane
// ends up with four consecutive columns of each of its rows: 16-byte stores, no transpose.  A wave owns one 64-column group.
template <int RT>
__device__ __forceinline__ void ch_dx_layer(const ChLayer& L, const float* in, float* out_lds, float* gdst, const int64_t ldg, const bool gadd,
                                            const float* mask, const int64_t ldmask, const int64_t row0, const int64_t batch,
                                            const int wave, const int c16, const int q) {
  const int K = L.out, N = L.in;
  const int Sin = ch_stride(K), Sout = ch_stride(N);
  const int groups = (N + 63) >> 6, kgs = (K + 15) >> 4;
  if (wave >= groups) return;           // (wave-uniform; the caller's barrier follows)
  constexpr int D = 4;
  const int col = wave * 64 + 4 * c16;  // this lane's four columns
  const bool colok = col < N;           // N % 4 == 0 (the host checks)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.w), 0, L.w_bytes, 0x00020000);
  auto loadB = [&](int kg, int e) -> f32x4 {
    const int k = kg * 16 + 4 * q + e;
    const unsigned off = (colok && k < K) ? (unsigned)(k * L.ldw + col) * 4u : CH_OOB;
    return bload4(rs, off);
  };
  // the relu' operand (the layer below's output): fetched up front, it has the whole k loop to arrive
  f32x4 mk[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t r = row0 + rt * 16 + 4 * q + i;
      mk[rt][i] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (mask && colok && r < batch) mk[rt][i] = *reinterpret_cast<const f32x4*>(mask + r * ldmask + col);
    }
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 Bf[D][4];
#pragma unroll
  for (int d = 0; d < D; d++)
#pragma unroll
    for (int e = 0; e < 4; e++) Bf[d][e] = loadB(d, e);
  const float* arow = in + c16 * Sin + 4 * q;
  auto step = [&](int kg, int d, bool refill) {
    f32x4 A[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) A[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 16 * Sin + kg * 16);
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt][t] = mfma4(A[rt][e], Bf[d][e][t], acc[rt][t]);
    if (refill) {
#pragma unroll
      for (int e = 0; e < 4; e++) Bf[d][e] = loadB(kg + D, e);
    }
    CH_PIN();
  };
  int kg0 = 0;
  for (; kg0 + D <= kgs; kg0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) step(kg0 + d, d, true);
  }
#pragma unroll
  for (int d = 0; d < D; d++)
    if (kg0 + d < kgs) step(kg0 + d, d, false);
  const int Np = (N + 15) & ~15;
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = rt * 16 + 4 * q + i;
      f32x4 v = f32x4{acc[rt][0][i], acc[rt][1][i], acc[rt][2][i], acc[rt][3][i]};
      const f32x4 m4 = mk[rt][i];
      v.x = m4.x > 0.0f ? v.x : 0.0f; v.y = m4.y > 0.0f ? v.y : 0.0f; v.z = m4.z > 0.0f ? v.z : 0.0f; v.w = m4.w > 0.0f ? v.w : 0.0f;
      if (gadd && colok && row0 + r < batch) v += *reinterpret_cast<const f32x4*>(gdst + (row0 + r) * ldg + col);      // (the accumulating form: the chain input has other consumers)
      if (!colok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (out_lds && col < Np) *reinterpret_cast<f32x4*>(out_lds + r * Sout + col) = v;
      if (colok && row0 + r < batch) *reinterpret_cast<f32x4*>(gdst + (row0 + r) * ldg + col) = v;
    }
}

template <int RT>
__global__ __launch_bounds__(CH_THREADS) void mlp_chain_dx_kernel(const ChainDxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  ffh_kernel_prio();
  constexpr int R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  float* const buf0 = ch_lds;
  float* const buf1 = ch_lds + a.buf1_off;
  {     // the top layer's dy rows -> LDS; a live activation derivative is applied here and written back in place, as the reference's
        // backward leaves dy [ref: src/ops/linear.cu:624-635]
    const ChLayer& T = a.L[a.n - 1];
    const int K = T.out, S = ch_stride(K), Kp = (K + 15) & ~15, k4 = Kp >> 2;
    for (int idx = tid; idx < R * k4; idx += CH_THREADS) {
      const int r = idx / k4, c = (idx - r * k4) * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row0 + r < a.batch && c < K) {
        float* p = T.dy + (row0 + r) * T.lddy + c;
        v = *reinterpret_cast<const f32x4*>(p);
        if (a.top_live) {
          const f32x4 y4 = *reinterpret_cast<const f32x4*>(T.y + (row0 + r) * T.ldy + c);
          if (T.act == FFH_AC_MODE_RELU) {
            v.x = y4.x > 0.0f ? v.x : 0.0f; v.y = y4.y > 0.0f ? v.y : 0.0f; v.z = y4.z > 0.0f ? v.z : 0.0f; v.w = y4.w > 0.0f ? v.w : 0.0f;
          } else {      // sigmoid [ref: src/ops/linear.cu:600-607]
            v.x = v.x * y4.x * (1.0f - y4.x); v.y = v.y * y4.y * (1.0f - y4.y); v.z = v.z * y4.z * (1.0f - y4.z); v.w = v.w * y4.w * (1.0f - y4.w);
          }
          *reinterpret_cast<f32x4*>(p) = v;
        }
      }
      *reinterpret_cast<f32x4*>(buf0 + r * S + c) = v;
    }
  }
  __syncthreads();
  const int lmin = a.dx ? 0 : 1;
  int step = 0;
  for (int l = a.n - 1; l >= lmin; l--, step++) {
    const ChLayer& L = a.L[l];
    const float* in = (step & 1) ? buf1 : buf0;
    float* out = (step & 1) ? buf0 : buf1;
    if (l > 0) {
      const ChLayer& B = a.L[l - 1];
      ch_dx_layer<RT>(L, in, out, B.dy, B.lddy, false, B.act == FFH_AC_MODE_RELU ? B.y : nullptr, B.ldy, row0, a.batch, wave, c16, q);
    } else {
      ch_dx_layer<RT>(L, in, nullptr, a.dx, a.lddx, a.dx_add != 0, a.xmask, a.ldxmask, row0, a.batch, wave, c16, q);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight / bias gradients: dW_l[m][n] += sum_b dy_l[b][m] x_l[b][n].  Both operands lie rows-are-k: lane (c, q) loads
// dy[b0 + q][m0 + 4c .. + 3] and x[b0 + q][n0 + 4c .. + 3] -- one k-step of four row tiles and four column tiles: two 16-byte loads
// feed 16 MFMAs.  Workgroup = (64 x 64 block of one layer's dW, batch split); its eight waves take the split's 4-row steps in turn,
// leave their blocks in LDS, and the workgroup adds the sum to dW with one atomic instruction per 256 contiguous bytes.
template <bool VEC>
__device__ __forceinline__ f32x4 dw_load(const __amdgpu_buffer_rsrc_t rs, const unsigned rowoff, const int col, const int lim) {
  if (VEC) return bload4(rs, (rowoff != CH_OOB && col < lim) ? rowoff + (unsigned)col * 4u : CH_OOB);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; e++) v[e] = bload1(rs, (rowoff != CH_OOB && col + e < lim) ? rowoff + (unsigned)(col + e) * 4u : CH_OOB);
  return v;
}

template <bool AVEC, bool BVEC>
__device__ __forceinline__ void ch_dw_block(const DwLayer& L, const int m0, const int n0, const int64_t r0, const int64_t r1, float* lds,
                                            const int tid, const int wave, const int c16, const int q, float* slot) {
  constexpr int D = 8;
  const int64_t nrows = r1 - r0;
  const int steps = (int)((nrows + 3) >> 2);              // 4-row k-steps of the split; wave w takes w, w + 8, ...
  const float* abase = L.dy + r0 * L.lddy;
  const float* bbase = L.x + r0 * L.ldx;
  auto extent = [](int64_t rows, int64_t ld, int width) -> unsigned {
    const int64_t b = ((rows - 1) * ld + width) * 4;
    return (unsigned)(b < 0x7fffffffLL ? b : 0x7fffffffLL);
  };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, extent(nrows, L.lddy, L.M), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bbase), 0, extent(nrows, L.ldx, L.N), 0x00020000);
  const int am = m0 + 4 * c16, bn = n0 + 4 * c16;
  auto rowoff = [&](int s, int64_t ld) -> unsigned {
    const int64_t r = (int64_t)s * 4 + q;
    return (s < steps && r < nrows) ? (unsigned)(r * ld * 4) : CH_OOB;
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_db = L.db != nullptr && n0 == 0;      // uniform
  f32x4 Af[D], Bf[D];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int s = wave + 8 * d;
    Af[d] = dw_load<AVEC>(rsA, rowoff(s, L.lddy), am, L.M);
    Bf[d] = dw_load<BVEC>(rsB, rowoff(s, L.ldx), bn, L.N);
  }
  const int mine = steps > wave ? (steps - wave + 7) >> 3 : 0;     // k-steps of this wave
  auto body = [&](int i, int d, bool refill) {
    const f32x4 av = Af[d], bv = Bf[d];
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 4; tn++) acc[tm][tn] = mfma4(av[tm], bv[tn], acc[tm][tn]);
    if (want_db) bsum += av;
    if (refill) {
      const int s = wave + 8 * (i + D);
      Af[d] = dw_load<AVEC>(rsA, rowoff(s, L.lddy), am, L.M);
      Bf[d] = dw_load<BVEC>(rsB, rowoff(s, L.ldx), bn, L.N);
    }
    CH_PIN();
  };
  int i0 = 0;
  for (; i0 + D <= mine; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) body(i0 + d, d, true);
  }
#pragma unroll
  for (int d = 0; d < D; d++)
    if (i0 + d < mine) body(i0 + d, d, false);
  // the wave's block -> its LDS slice, row-major 64 x 64: lane (c, q) holds, for tile pair (tm, tn) and i, the element
  // (16q + 4i + tm, 4c + tn): for fixed (tm, i) four consecutive columns
  float* slice = lds + wave * 4096;
#pragma unroll
  for (int tm = 0; tm < 4; tm++)
#pragma unroll
    for (int i = 0; i < 4; i++)
      *reinterpret_cast<f32x4*>(slice + (16 * q + 4 * i + tm) * 64 + 4 * c16) = f32x4{acc[tm][0][i], acc[tm][1][i], acc[tm][2][i], acc[tm][3][i]};
  float* dbl = lds + 8 * 4096;        // [8 waves][64]
  if (want_db) {
    f32x4 t = bsum;
#pragma unroll
    for (int e = 0; e < 4; e++) { t[e] += __shfl_xor(t[e], 16); t[e] += __shfl_xor(t[e], 32); }
    if (q == 0) *reinterpret_cast<f32x4*>(dbl + wave * 64 + 4 * c16) = t;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int e = tid + CH_THREADS * j;          // element of the block: row e / 64, column e % 64 (a wave = one row: 256 contiguous bytes)
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) v += lds[w * 4096 + e];
    const int m = m0 + (e >> 6), n = n0 + (e & 63);
    if (slot) slot[e] = v;                                   // deterministic mode: the block as it is (the sums above run in a fixed order)
    else if (m < L.M && n < L.N) atomicAdd(L.dw + (int64_t)m * L.ldw + n, v);
  }
  if (want_db && tid < 64) {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) v += dbl[w * 64 + tid];
    if (slot) slot[4096 + tid] = v;
    else if (m0 + tid < L.M) atomicAdd(L.db + m0 + tid, v);
  }
}

__global__ __launch_bounds__(CH_THREADS) void mlp_chain_dw_kernel(const ChainDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ch_lds[];
  ffh_kernel_prio();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int item = (int)(blockIdx.x / (unsigned)a.S), split = (int)(blockIdx.x % (unsigned)a.S);
  int l = 0;
  while (l + 1 < a.n && item >= a.L[l + 1].first_item) l++;
  const DwLayer& L = a.L[l];
  const int li = item - L.first_item;
  const int m0 = (li / L.nbn) * 64, n0 = (li % L.nbn) * 64;
  const int64_t r0 = (int64_t)split * a.rows_per_split;
  int64_t r1 = r0 + a.rows_per_split;
  if (r1 > a.batch) r1 = a.batch;
  if (r0 >= r1) return;
  float* slot = a.slots ? a.slots + (size_t)blockIdx.x * kSkinnyWsRow : nullptr;
  if (L.a_vec) {
    if (L.b_vec) ch_dw_block<true, true>(L, m0, n0, r0, r1, ch_lds, tid, wave, c16, q, slot);
    else ch_dw_block<true, false>(L, m0, n0, r0, r1, ch_lds, tid, wave, c16, q, slot);
  } else {
    if (L.b_vec) ch_dw_block<false, true>(L, m0, n0, r0, r1, ch_lds, tid, wave, c16, q, slot);
    else ch_dw_block<false, false>(L, m0, n0, r0, r1, ch_lds, tid, wave, c16, q, slot);
  }
}

// Deterministic mode (ffh_ctx_set_deterministic): one workgroup per 64 x 64 block of some layer's dW adds the block's batch splits in split
// order -- the one writer of those elements of dW / db, so the result does not depend on which workgroup finished first (the SPLIT /
// skinny pattern, as its own launch: no arrival protocol).  Splits beyond the batch wrote nothing and are skipped, as the dW kernel skipped them.
__global__ __launch_bounds__(CH_THREADS) void mlp_chain_dw_reduce_kernel(const ChainDwArgs a) {
  ffh_kernel_prio();
  const int tid = threadIdx.x, item = (int)blockIdx.x;
  int l = 0;
  while (l + 1 < a.n && item >= a.L[l + 1].first_item) l++;
  const DwLayer& L = a.L[l];
  const int li = item - L.first_item;
  const int m0 = (li / L.nbn) * 64, n0 = (li % L.nbn) * 64;
  int S = 0;
  while (S < a.S && (int64_t)S * a.rows_per_split < a.batch) S++;
  const float* slots = a.slots + (size_t)item * a.S * kSkinnyWsRow;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int e = tid + CH_THREADS * j;
    const int m = m0 + (e >> 6), n = n0 + (e & 63);
    if (m >= L.M || n >= L.N) continue;
    float v = 0.0f;
    for (int sp = 0; sp < S; sp++) v += slots[(size_t)sp * kSkinnyWsRow + e];
    float* d = L.dw + (int64_t)m * L.ldw + n;
    *d = *d + v;
  }
  if (L.db && n0 == 0 && tid < 64 && m0 + tid < L.M) {
    float v = 0.0f;
    for (int sp = 0; sp < S; sp++) v += slots[(size_t)sp * kSkinnyWsRow + 4096 + tid];
    L.db[m0 + tid] += v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename K>
int ch_set_lds(ffh_ctx* c, K kern, int bytes, signed char* ok) {
  const int dev = c->device & 63;
  if (ok[dev] == 0) ok[dev] = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) ? 1 : -1;
  if (ok[dev] < 0) { (void)hipGetLastError(); return 0; }
  return 1;
}
constexpr int CH_LDS_MAX = 150 * 1024;

const char* chain_check(const ffh_chain_layer* ls, int n, int64_t batch) {
  if (!ls || n < 1 || n > CH_MAXL) return "layer count";
  if (batch < 0 || batch >= (1LL << 31)) return "batch";
  for (int l = 0; l < n; l++) {
    const ffh_chain_layer& L = ls[l];
    if (L.in_dim < 1 || L.out_dim < 1 || L.in_dim > CH_MAXW || L.out_dim > CH_MAXW) return "layer width (1 .. 512)";
    if (L.ldw < L.in_dim || L.ldy < L.out_dim) return "leading dimension";
    if (!L.w || !L.y) return "null pointer";
    if (l > 0 && L.in_dim != ls[l - 1].out_dim) return "widths do not chain";
  }
  return nullptr;
}

void fill_layer(ChLayer& d, const ffh_chain_layer& s) {
  d.w = s.w; d.bias = s.bias; d.y = s.y; d.dy = s.dy; d.dw = s.dw; d.db = s.db;
  d.ldy = s.ldy; d.lddy = s.lddy; d.ldw = s.ldw; d.in = s.in_dim; d.out = s.out_dim; d.act = s.activation;
  d.w_bytes = (unsigned)(((int64_t)(s.out_dim - 1) * s.ldw + s.in_dim) * 4);
  d.w_vec = (al16(s.w) && (s.ldw & 3) == 0 && (s.in_dim & 3) == 0) ? 1 : 0;
}

}  // namespace

// The chains run on the exact-fp32 kernels.  FFH_MATH_FP32_SPLIT_BF16X3 proper leaves a layer to those kernels where its GEMMs are below
// FFH_BF16X3_MIN_FLOP (round 6: a chain of narrow layers always is, up to ~19000 samples at 512 x 512) -- then a chain is the per-layer calls of
// that mode exactly; a chain with a layer the mode would take, the tensor-op mode and the every-shape test mode refuse the chain as before.
static bool chain_math_mode_ok(const ffh_ctx* c, const ffh_chain_layer* layers, int nlayers, int64_t batch) {
  if (c->math_mode == FFH_MATH_DEFAULT) return true;
  if (c->math_mode != FFH_MATH_FP32_SPLIT_BF16X3) return false;
  for (int l = 0; l < nlayers; l++)
    if (ffh_gemm::use_bf16(c, layers[l].in_dim, layers[l].out_dim, batch)) return false;
  return true;
}

extern "C" {

int ffh_mlp_chain_fwd(ffh_ctx* c, const float* x, int64_t ldx, const ffh_chain_layer* layers, int nlayers, int64_t batch, ffh_stream s) {
  const char* bad = chain_check(layers, nlayers, batch);
  if (bad) { char m[96]; snprintf(m, sizeof m, "mlp_chain_fwd: %s", bad); return ffh_fail(c, FFH_ERR_BAD_ARG, m); }
  FFH_REQUIRE(c, batch == 0 || x, "mlp_chain_fwd: null pointer");
  FFH_REQUIRE(c, ldx >= layers[0].in_dim, "mlp_chain_fwd: leading dimension");
  for (int l = 0; l < nlayers; l++) {
    const int act = layers[l].activation;
    if (act != FFH_AC_MODE_NONE && act != FFH_AC_MODE_RELU && act != FFH_AC_MODE_SIGMOID && act != FFH_AC_MODE_GELU)
      return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: activation not supported (NONE, RELU, SIGMOID, GELU)");
  }
  if (!chain_math_mode_ok(c, layers, nlayers, batch)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: layers of the exact-fp32 kernels only (math mode)");
  ffh_route_clear(c);
  if (batch == 0) return FFH_OK;
  ChainFwdArgs a{};
  a.x = x; a.ldx = ldx; a.batch = batch; a.n = nlayers;
  a.x_vec = (al16(x) && (ldx & 3) == 0 && (layers[0].in_dim & 3) == 0) ? 1 : 0;
  for (int l = 0; l < nlayers; l++) fill_layer(a.L[l], layers[l]);
  int w0 = 0, w1 = 0;     // widest block each buffer holds: position p (input of layer p; p = n: the last output) lives in buffer p & 1
  for (int p = 0; p <= nlayers; p++) {
    const int w = ch_stride(p == 0 ? layers[0].in_dim : layers[p - 1].out_dim);
    if (p & 1) { if (w > w1) w1 = w; } else { if (w > w0) w0 = w; }
  }
  int rt = batch >= 32LL * c->num_cus ? 2 : 1;
  if (rt == 2 && (32 * (w0 + w1) + 8 * CH_STAGE_FLOATS) * 4 > CH_LDS_MAX) rt = 1;      // (two 512-wide blocks of 32 rows do not fit beside the weight images)
  const int R = 16 * rt;
  a.buf1_off = R * w0;
  a.stage_off = R * (w0 + w1);
  const int lds = (R * (w0 + w1) + 8 * CH_STAGE_FLOATS) * 4;
  if (lds > CH_LDS_MAX) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: LDS");
  const unsigned grid = (unsigned)((batch + R - 1) / R);
  static signed char ok1[64], ok2[64];
  if (rt == 1) {
    if (!ch_set_lds(c, mlp_chain_fwd_kernel<1>, CH_LDS_MAX, ok1)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: LDS attribute");
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<1>, dim3(grid), dim3(CH_THREADS), lds, as_stream(s), a);
  } else {
    if (!ch_set_lds(c, mlp_chain_fwd_kernel<2>, CH_LDS_MAX, ok2)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_fwd: LDS attribute");
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<2>, dim3(grid), dim3(CH_THREADS), lds, as_stream(s), a);
  }
  FFH_LAUNCH_CHECK(c, "mlp_chain_fwd_kernel");
  { char tok[64]; snprintf(tok, sizeof tok, "mlp_chain_fwd|layers=%d|rows=%d", nlayers, R); ffh_route_add(c, tok); }
  return FFH_OK;
}

int ffh_mlp_chain_bwd(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const ffh_chain_layer* layers, int nlayers,
                      int64_t batch, int flags, ffh_stream s) {
  const char* bad = chain_check(layers, nlayers, batch);
  if (bad) { char m[96]; snprintf(m, sizeof m, "mlp_chain_bwd: %s", bad); return ffh_fail(c, FFH_ERR_BAD_ARG, m); }
  FFH_REQUIRE(c, batch == 0 || x, "mlp_chain_bwd: null pointer");
  FFH_REQUIRE(c, ldx >= layers[0].in_dim && (!dx || lddx >= layers[0].in_dim), "mlp_chain_bwd: leading dimension");
  FFH_REQUIRE(c, (flags & ~(FFH_LINEAR_DX_OVERWRITE | FFH_LINEAR_DX_MASK_BY_X | FFH_LINEAR_DY_PREMASKED)) == 0, "mlp_chain_bwd: flags");
  for (int l = 0; l < nlayers; l++) {
    FFH_REQUIRE(c, layers[l].dy && layers[l].dw && layers[l].lddy >= layers[l].out_dim, "mlp_chain_bwd: null pointer / leading dimension");
  }
  if (!chain_math_mode_ok(c, layers, nlayers, batch)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: layers of the exact-fp32 kernels only (math mode)");
  const ffh_chain_layer& top = layers[nlayers - 1];
  const bool premasked = (flags & FFH_LINEAR_DY_PREMASKED) != 0;
  if (top.activation != FFH_AC_MODE_NONE && top.activation != FFH_AC_MODE_RELU && top.activation != FFH_AC_MODE_SIGMOID)
    return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: activation not supported (NONE, RELU, SIGMOID; GELU is forward-only, as in the reference)");
  for (int l = 0; l + 1 < nlayers; l++)
    if (layers[l].activation != FFH_AC_MODE_NONE && layers[l].activation != FFH_AC_MODE_RELU)
      return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: inner layers NONE or RELU");
  // the data-gradient kernel wants 16-byte rows everywhere it stores or masks
  for (int l = 0; l < nlayers; l++) {
    const ffh_chain_layer& L = layers[l];
    const bool dx_here = l > 0 || dx != nullptr;
    if (dx_here && ((L.in_dim & 3) || (L.ldw & 3) || !al16(L.w))) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: in_dim / ldw not multiples of 4");
    if ((L.out_dim & 3) && l == nlayers - 1) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: top out_dim not a multiple of 4");
    if (!al16(L.dy) || (L.lddy & 3) || !al16(L.y) || (L.ldy & 3)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: y / dy rows not 16-byte aligned");
  }
  if (dx && (!al16(dx) || (lddx & 3))) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: dx rows not 16-byte aligned");
  if (dx && (flags & FFH_LINEAR_DX_MASK_BY_X) && (!al16(x) || (ldx & 3))) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: x rows not 16-byte aligned");
  ffh_route_clear(c);
  if (batch == 0) return FFH_OK;

  // ---- 0. everything that can still say "not served" -- BEFORE the first launch: the data-gradient kernel masks dy in place and stores or
  //         accumulates dx, so a later FFH_ERR_UNSUPPORTED would have the caller's per-layer calls do that a second time (round-5 advisor) ----
  ChainDwArgs w{};
  w.batch = batch; w.n = nlayers;
  int items = 0;
  for (int l = 0; l < nlayers; l++) {
    const ffh_chain_layer& L = layers[l];
    DwLayer& d = w.L[l];
    d.dy = L.dy; d.lddy = L.lddy;
    d.x = l == 0 ? x : layers[l - 1].y; d.ldx = l == 0 ? ldx : layers[l - 1].ldy;
    d.dw = L.dw; d.db = L.db; d.ldw = L.ldw; d.M = L.out_dim; d.N = L.in_dim;
    d.nbn = (L.in_dim + 63) / 64;
    d.first_item = items;
    items += ((L.out_dim + 63) / 64) * d.nbn;
    d.a_vec = (al16(d.dy) && (d.lddy & 3) == 0 && (d.M & 3) == 0) ? 1 : 0;
    d.b_vec = (al16(d.x) && (d.ldx & 3) == 0 && (d.N & 3) == 0) ? 1 : 0;
  }
  w.nitems = items;
  int S = c->num_cus / items;
  if (S < 1) S = 1;
  const int64_t max_s = (batch + 63) / 64;                 // at least two 4-row steps per wave
  if (S > max_s) S = (int)max_s;
  int64_t rps = (batch + S - 1) / S;
  rps = (rps + 31) / 32 * 32;
  S = (int)((batch + rps - 1) / rps);
  w.S = S; w.rows_per_split = rps;
  if (c->deterministic) {
    // the splits' blocks meet in the stream's scratch (ffh_ctx_reserve_scratch: the narrow-layer backward's partial rows, one row per
    // workgroup here) and a second launch adds them in split order
    for (int i = 0; i < c->nscratch; i++)
      if (c->scratch[i].stream == (void*)as_stream(s) && c->scratch[i].skinny_ws) w.slots = c->scratch[i].skinny_ws;
    if (!w.slots || (int64_t)items * S > kSkinnyWsBlocks)
      return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: deterministic mode needs the stream's scratch (ffh_ctx_reserve_scratch)");
  }
  const int lds_dw = (8 * 4096 + 8 * 64) * 4;
  static signed char okw[64];
  if (!ch_set_lds(c, mlp_chain_dw_kernel, lds_dw, okw)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: LDS attribute");

  // ---- 1. the data-gradient chain (nothing to do for a single layer whose input gradient is discarded and whose dy is final) ----
  const bool top_live = !premasked && top.activation != FFH_AC_MODE_NONE;
  int rt = batch >= 32LL * c->num_cus ? 2 : 1;
  {
    int v0 = 0, v1 = 0;
    for (int j = 0; j < nlayers; j++) { const int w = ch_stride(layers[nlayers - 1 - j].out_dim); if (j & 1) { if (w > v1) v1 = w; } else { if (w > v0) v0 = w; } }
    if (rt == 2 && 32 * (v0 + v1) * 4 > CH_LDS_MAX) rt = 1;
  }
  const int R = 16 * rt;
  if (nlayers > 1 || dx || top_live) {
    ChainDxArgs a{};
    a.dx = dx; a.lddx = lddx; a.batch = batch; a.n = nlayers;
    a.xmask = (dx && (flags & FFH_LINEAR_DX_MASK_BY_X)) ? x : nullptr; a.ldxmask = ldx;
    a.dx_add = (flags & FFH_LINEAR_DX_OVERWRITE) ? 0 : 1;
    a.top_live = top_live ? 1 : 0;
    for (int l = 0; l < nlayers; l++) fill_layer(a.L[l], layers[l]);
    int w0 = 0, w1 = 0;   // position j: the dy of layer n - 1 - j
    for (int j = 0; j < nlayers; j++) {
      const int w = ch_stride(layers[nlayers - 1 - j].out_dim);
      if (j & 1) { if (w > w1) w1 = w; } else { if (w > w0) w0 = w; }
    }
    a.buf1_off = R * w0;
    const int lds = R * (w0 + w1) * 4;
    if (lds > CH_LDS_MAX) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: LDS");
    const unsigned grid = (unsigned)((batch + R - 1) / R);
    static signed char ok1[64], ok2[64];
    if (rt == 1) {
      if (!ch_set_lds(c, mlp_chain_dx_kernel<1>, CH_LDS_MAX, ok1)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: LDS attribute");
      hipLaunchKernelGGL(mlp_chain_dx_kernel<1>, dim3(grid), dim3(CH_THREADS), lds, as_stream(s), a);
    } else {
      if (!ch_set_lds(c, mlp_chain_dx_kernel<2>, CH_LDS_MAX, ok2)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mlp_chain_bwd: LDS attribute");
      hipLaunchKernelGGL(mlp_chain_dx_kernel<2>, dim3(grid), dim3(CH_THREADS), lds, as_stream(s), a);
    }
    FFH_LAUNCH_CHECK(c, "mlp_chain_dx_kernel");
    { char tok[64]; snprintf(tok, sizeof tok, "mlp_chain_dx|layers=%d|rows=%d", nlayers, R); ffh_route_add(c, tok); }
  }
  // ffh_event_record_with_next_linear_bwd: the chain input's gradient is complete here, in front of the weight gradients
  if (c->attach_event) {
    hipEvent_t ev = (hipEvent_t)c->attach_event;
    c->attach_event = nullptr;
    FFH_HIP_TRY(c, hipEventRecord(ev, as_stream(s)));
  }

  // ---- 2. every layer's weight / bias gradient ----
  hipLaunchKernelGGL(mlp_chain_dw_kernel, dim3((unsigned)(items * S)), dim3(CH_THREADS), lds_dw, as_stream(s), w);
  FFH_LAUNCH_CHECK(c, "mlp_chain_dw_kernel");
  if (w.slots) {
    hipLaunchKernelGGL(mlp_chain_dw_reduce_kernel, dim3((unsigned)items), dim3(CH_THREADS), 0, as_stream(s), w);
    FFH_LAUNCH_CHECK(c, "mlp_chain_dw_reduce_kernel");
  }
  { char tok[80]; snprintf(tok, sizeof tok, "mlp_chain_dw|blocks=%d|splits=%d%s", items, S, w.slots ? "|ordered" : ""); ffh_route_add(c, tok); }
  return FFH_OK;
}

}  // extern "C"
