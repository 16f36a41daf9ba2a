// gemm_big_lab.hip -- stand-alone bench for the BIG fp32 MFMA GEMMs of the Terabyte / MLPerf step (not part of the product).
//   forward form: C[m][n] = relu(sum_k A[m][k] * B[n][k] + bias[n]), both operands k-contiguous.
// Candidate "direct": no LDS and no barrier at all.  Every wave streams its own MFMA operand fragments from global
// memory into registers as 16-byte pieces: lane (row r, half h) of v_mfma_f32_32x32x2_f32 may take any k as long as both
// operands agree, so it loads KG/2 consecutive floats of its row (k = kg + (KG/2) h + j) and the j-th MFMA of the group
// multiplies the pair {kg + j, kg + KG/2 + j}.  A register ring of NBUF groups keeps the loads ahead of the MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_big_lab.hip -o tools/lab/gemm_big_lab
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// XCD-aware tile order: consecutive tiles (same m rows, all n) land on one XCD
__device__ __forceinline__ void tile_of_block(int& bx, int& by, unsigned nbx, unsigned nby) {
  const unsigned total = nbx * nby;
  const unsigned lin = blockIdx.x;
  const unsigned xcd = lin & 7u, loc = lin >> 3, q = total >> 3, rem = total & 7u;
  const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
  bx = (int)(nlin % nbx); by = (int)(nlin / nbx);
}

// WM x WN MFMA tiles (32x32) per wave; 2x2 waves per block; KG = k per group (16 or 32); NBUF = groups in flight
template <int WM, int WN, int KG, int NBUF>
__global__ __launch_bounds__(256) void gemm_direct(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   const float* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                                   int relu) {
  constexpr int Q = KG / 8;              // float4 per lane per tile per group
  constexpr int BM = 2 * WM * 32, BN = 2 * WN * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by;
  tile_of_block(bx, by, (N + BN - 1) / BN, (M + BM - 1) / BM);
  const int m0 = by * BM + (wave >> 1) * (WM * 32), n0 = bx * BN + (wave & 1) * (WN * 32);
  const float* ap[WM];
  const float* bp[WN];
#pragma unroll
  for (int i = 0; i < WM; i++) ap[i] = A + (int64_t)(m0 + i * 32 + lr) * lda + (KG / 2) * lh;
#pragma unroll
  for (int j = 0; j < WN; j++) bp[j] = B + (int64_t)(n0 + j * 32 + lr) * ldb + (KG / 2) * lh;
  float4 ra[NBUF][WM][Q], rb[NBUF][WN][Q];
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  const int ng = K / KG;
  auto load = [&](int g, auto buf_tag) {
    constexpr int b = decltype(buf_tag)::value;
    const int k0 = g * KG;
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
      for (int q = 0; q < Q; q++) ra[b][i][q] = ld4(ap[i] + k0 + 4 * q);
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int q = 0; q < Q; q++) rb[b][j][q] = ld4(bp[j] + k0 + 4 * q);
  };
  auto compute = [&](auto buf_tag) {
    constexpr int b = decltype(buf_tag)::value;
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
          for (int j = 0; j < WN; j++) {
            const float a = e == 0 ? ra[b][i][q].x : e == 1 ? ra[b][i][q].y : e == 2 ? ra[b][i][q].z : ra[b][i][q].w;
            const float bb = e == 0 ? rb[b][j][q].x : e == 1 ? rb[b][j][q].y : e == 2 ? rb[b][j][q].z : rb[b][j][q].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i][j], 0, 0, 0);
          }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (NBUF == 2) {
    load(0, I0{});
    int g = 0;
    for (; g + 2 <= ng; g += 2) {
      load(g + 1, I1{});
      compute(I0{});
      if (g + 2 < ng) load(g + 2, I0{});
      compute(I1{});
    }
    if (g < ng) compute(I0{});
  } else {
    load(0, I0{});
    if (ng > 1) load(1, I1{});
    int g = 0;
    for (; g + 3 <= ng; g += 3) {
      if (g + 2 < ng) load(g + 2, I2{});
      compute(I0{});
      if (g + 3 < ng) load(g + 3, I0{});
      compute(I1{});
      if (g + 4 < ng) load(g + 4, I1{});
      compute(I2{});
    }
    if (g < ng) { if (g + 2 < ng) load(g + 2, I2{}); compute(I0{}); g++; }
    if (g < ng) { compute(I1{}); g++; }
    if (g < ng) { compute(I2{}); g++; }
  }
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int n = n0 + j * 32 + lr;
      const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[i][j][r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Candidate "dma": LDS-DMA staged (global_load_lds_dwordx4), NSTAGE k-tiles of BK in LDS, ONE barrier per k-tile, every
// wave owns WM x WN 32x32 tiles over the WHOLE k-tile (no k split inside the workgroup), operand fragments read by
// ds_read_b128 (k-contiguous image, XOR-swizzled on the source address) or ds_read_b32 (rows-are-k image) one k-octet
// ahead of the MFMAs that use them.
//   C[m][n] (op)= sum_k A(m,k) B(n,k);  AKR / BKR = false: rows of the operand are m / n, k contiguous; true: rows are k.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ unsigned long long g_clk[4];
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <bool AKR, bool BKR, int WGM, int WGN, int WM, int WN, int BK, int NSTAGE>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_dma(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          const float* __restrict__ bias, const float* __restrict__ zeros, int M, int N, int K,
                                                          int64_t lda, int64_t ldb, int64_t ldc, int relu) {
  constexpr int NW = WGM * WGN;
  constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
  constexpr int CPR = BK / 4;                       // 16-byte chunks per k-contiguous row
  constexpr int RPB = 16 / CPR;                     // rows per 256-byte bank row
  constexpr int A_CH = BM * CPR, B_CH = BN * CPR, STAGE_CH = A_CH + B_CH;
  constexpr int NPIECE = STAGE_CH / 64;
  constexpr int NIW = NPIECE / NW;
  constexpr int KG = BK / 8;                        // k-octets per k-tile
  static_assert(NPIECE % NW == 0, "pieces per wave");
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by;
  tile_of_block(bx, by, (N + BN - 1) / BN, (M + BM - 1) / BM);
  const int m0 = by * BM, n0 = bx * BN;
  const int wgm = wave / WGN, wgn = wave % WGN;
  const int wm0 = wgm * WM * 32, wn0 = wgn * WN * 32;

  const float* src[NIW];
  int src_k[NIW];
#pragma unroll
  for (int i = 0; i < NIW; i++) {
    const int q = wave + NW * i;
    const int ch = q * 64 + lane;
    const bool isA = ch < A_CH;
    const int cb = isA ? ch : ch - A_CH;
    const float* base = isA ? A : B;
    const int64_t ld = isA ? lda : ldb;
    const int o0 = isA ? m0 : n0, olim = isA ? M : N;
    const bool kr = isA ? AKR : BKR;
    const int cols4 = (isA ? BM : BN) / 4;
    if (!kr) {
      const int r = cb / CPR, c = (cb % CPR) ^ ((r / RPB) & (CPR - 1));
      src[i] = (o0 + r < olim) ? base + (int64_t)(o0 + r) * ld + 4 * c : nullptr;
      src_k[i] = 4 * c;
    } else {
      const int kr_ = cb / cols4, c = cb - kr_ * cols4;
      src[i] = (o0 + 4 * c < olim) ? base + (int64_t)kr_ * ld + o0 + 4 * c : nullptr;
      src_k[i] = kr_ | (1 << 30);
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
  auto issue = [&](int kt, int buf, int i0, int i1) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      const int q = wave + NW * i;
      const bool rows_k = (src_k[i] >> 30) & 1;
      const int kk = k0 + (src_k[i] & 0xFFFF);
      const int64_t ld = (q * 64 < A_CH) ? lda : ldb;
      const float* p = (src[i] != nullptr && kk < K) ? (rows_k ? src[i] + (int64_t)k0 * ld : src[i] + k0) : zeros;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE_CH + q * 64) * 16u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  const int nk = (K + BK - 1) / BK;
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) issue(s, s, 0, NIW);
  int buf = 0, nbuf = NSTAGE - 1;
  for (int t = 0; t < nk; t++) {
    wait_vmcnt<(NSTAGE - 2) * NIW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* as4 = lds + buf * STAGE_CH;
    const float4* bs4 = as4 + A_CH;
    const float* as1 = reinterpret_cast<const float*>(as4);
    const float* bs1 = reinterpret_cast<const float*>(bs4);
    auto rd = [&](int j, float4 (&a)[WM], float4 (&b)[WN]) {
#pragma unroll
      for (int i = 0; i < WM; i++) {
        const int ra = wm0 + i * 32 + lr;
        if (!AKR) a[i] = as4[ra * CPR + ((2 * j + lh) ^ ((ra / RPB) & (CPR - 1)))];
        else {
          const float* q = as1 + (8 * j + 4 * lh) * BM + ra;
          a[i] = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
        }
      }
#pragma unroll
      for (int i = 0; i < WN; i++) {
        const int rb = wn0 + i * 32 + lr;
        if (!BKR) b[i] = bs4[rb * CPR + ((2 * j + lh) ^ ((rb / RPB) & (CPR - 1)))];
        else {
          const float* q = bs1 + (8 * j + 4 * lh) * BN + rb;
          b[i] = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
        }
      }
    };
    float4 a_cur[WM], b_cur[WN], a_nxt[WM], b_nxt[WN];
    rd(0, a_cur, b_cur);
#pragma unroll
    for (int j = 0; j < KG; j++) {
      if (j + 1 < KG) rd(j + 1, a_nxt, b_nxt);
      issue(t + NSTAGE - 1, nbuf, j * NIW / KG, (j + 1) * NIW / KG);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
          for (int jn = 0; jn < WN; jn++) {
            const float av = e == 0 ? a_cur[i].x : e == 1 ? a_cur[i].y : e == 2 ? a_cur[i].z : a_cur[i].w;
            const float bv = e == 0 ? b_cur[jn].x : e == 1 ? b_cur[jn].y : e == 2 ? b_cur[jn].z : b_cur[jn].w;
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jn], 0, 0, 0);
          }
#pragma unroll
      for (int i = 0; i < WM; i++) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int i = 0; i < WN; i++) b_cur[i] = b_nxt[i];
    }
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }
  wait_vmcnt<0>();
  if (blockIdx.x == 0 && tid == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Candidate "plr": register-staged (global_load_dwordx4 -> ds_write_b128, no transpose: the LDS image is k-contiguous and
// XOR-swizzled), THREE LDS buffers so that the fragments of the next k-tile's first k-octet can be read BEFORE the barrier
// that ends the current k-tile: the MFMA stream of a wave never waits for an LDS round trip, and the barrier only has to
// absorb wave skew.  One barrier per k-tile.  Forward form (both operands k-contiguous), interior tiles only (lab).
template <int BK, int WM, int WN, int ABL = 0>   // ABL 1: every k-tile re-reads tile 0 (operands stay in L1 / L2); 2: no global loads in the loop
__global__ __launch_bounds__(256) void gemm_plr(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                const float* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int relu) {
  constexpr int BM = 2 * WM * 32, BN = 2 * WN * 32;
  constexpr int CPR = BK / 4, RPB = 16 / CPR;
  constexpr int A_CH = BM * CPR, B_CH = BN * CPR, STAGE_CH = A_CH + B_CH;
  constexpr int NA = A_CH / 256, NB = B_CH / 256;        // float4 per thread per operand per k-tile
  constexpr int KG = BK / 8;
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by;
  tile_of_block(bx, by, N / BN, M / BM);
  const int m0 = by * BM, n0 = bx * BN;
  const int wm0 = (wave >> 1) * WM * 32, wn0 = (wave & 1) * WN * 32;
  // staging roles
  const int sc = tid % CPR, sr = tid / CPR;              // chunk, row (+ i * 256 / CPR)
  constexpr int RPP = 256 / CPR;
  const float* ga[NA];
  const float* gb[NB];
  int wa[NA], wb[NB];
#pragma unroll
  for (int i = 0; i < NA; i++) { const int r = sr + i * RPP; ga[i] = A + (int64_t)(m0 + r) * lda + 4 * sc; wa[i] = r * CPR + (sc ^ ((r / RPB) & (CPR - 1))); }
#pragma unroll
  for (int i = 0; i < NB; i++) { const int r = sr + i * RPP; gb[i] = B + (int64_t)(n0 + r) * ldb + 4 * sc; wb[i] = A_CH + r * CPR + (sc ^ ((r / RPB) & (CPR - 1))); }
  float4 ra[NA], rb[NB];
  auto gload = [&](int kt0) {
    const int kt = ABL == 1 ? (kt0 & 1) : kt0;
    if (ABL == 2 && kt0 > 2) return;
#pragma unroll
    for (int i = 0; i < NA; i++) ra[i] = ld4(ga[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < NB; i++) rb[i] = ld4(gb[i] + kt * BK);
  };
  auto lwrite = [&](int buf) {
    float4* d = lds + buf * STAGE_CH;
#pragma unroll
    for (int i = 0; i < NA; i++) d[wa[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < NB; i++) d[wb[i]] = rb[i];
  };
  // fragment addresses (chunk index inside a stage) for k-octet j: row * CPR + ((2j + lh) ^ sw(row))
  int fa[WM], fb[WN], sa[WM], sb[WN];
#pragma unroll
  for (int i = 0; i < WM; i++) { const int r = wm0 + i * 32 + lr; fa[i] = r * CPR; sa[i] = (r / RPB) & (CPR - 1); }
#pragma unroll
  for (int i = 0; i < WN; i++) { const int r = wn0 + i * 32 + lr; fb[i] = A_CH + r * CPR; sb[i] = (r / RPB) & (CPR - 1); }
  auto rd = [&](int buf, int j, float4 (&a)[WM], float4 (&b)[WN]) {
    const float4* s4 = lds + buf * STAGE_CH;
#pragma unroll
    for (int i = 0; i < WM; i++) a[i] = s4[fa[i] + ((2 * j + lh) ^ sa[i])];
#pragma unroll
    for (int i = 0; i < WN; i++) b[i] = s4[fb[i] + ((2 * j + lh) ^ sb[i])];
  };
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  auto mfmas = [&](const float4 (&a)[WM], const float4 (&b)[WN]) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int jn = 0; jn < WN; jn++) {
          const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
          const float bv = e == 0 ? b[jn].x : e == 1 ? b[jn].y : e == 2 ? b[jn].z : b[jn].w;
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jn], 0, 0, 0);
        }
  };
  const int nk = K / BK;
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // prologue: tiles 0 and 1 in LDS, tile 2 in registers (tile indices beyond the last are clamped: the extra loads are
  // harmless and keep the loop body free of branches)
  const int last = nk - 1;
  gload(0); lwrite(0);
  gload(1 < last ? 1 : last); lwrite(1);
  gload(2 < last ? 2 : last);
  __syncthreads();
  float4 a0[WM], b0[WN], a1[WM], b1[WN];
  rd(0, 0, a0, b0);
  __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0): the loop is entered with nothing pending
  int b_cur = 0, b_nxt = 1, b_wr = 2;
  for (int t = 0; t < nk; t++) {
#pragma unroll
    for (int j = 0; j < KG; j += 2) {
      // even octet j: operands in (a0, b0); fetch octet j + 1 into (a1, b1)
      if (j + 1 < KG) rd(b_cur, j + 1, a1, b1); else rd(b_nxt, 0, a1, b1);
      if (j == 0) lwrite(b_wr);                                     // tile t + 2: registers -> LDS (loaded one iteration ago)
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) gload(t + 3 < last ? t + 3 : last);               // tile t + 3 -> registers
      if (j + 1 < KG) {
        if (j + 2 < KG) rd(b_cur, j + 2, a0, b0); else rd(b_nxt, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the LDS writes of this iteration (older than the last WM + WN fragment reads) must have landed; those reads may fly on
    __builtin_amdgcn_s_waitcnt(0xC07F | ((WM + WN) << 8) & 0x0F00);
    __builtin_amdgcn_s_barrier();
    const int tmp = b_cur; b_cur = b_nxt; b_nxt = b_wr; b_wr = tmp;
  }
  if (blockIdx.x == 0 && tid == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[i][j][r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
}

// pure MFMA loops on random register operands: what the matrix pipe delivers at the clock the chip holds
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_only(const float* __restrict__ src, float* __restrict__ dst, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = src[(tid * 16 + i) & 0xFFFFF]; b[i] = src[(tid * 16 + 8 + i) & 0xFFFFF]; }
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 7], b[(k + 2 * i) & 7], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) sum += acc[i][r];
  } else {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 7], b[(k + (i >> 2)) & 7], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) sum += acc[i][r];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  dst[tid] = sum;
}

template <typename F>
float time_it(F f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 3456;
  const int iters = 10;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N), hC((size_t)M * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.0f;
  float *A, *B, *C, *bias;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, hC.size() * 4)); CK(hipMalloc(&bias, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  auto check = [&](const char* name, float us) {
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 4000; t++) {
      const int m = (int)(((int64_t)t * 7919) % M), n = (int)(((int64_t)t * 104729) % N);
      double s = hb[n];
      for (int k = 0; k < K; k++) s += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
      if (s < 0) s = 0;
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)m * N + n]) / (1.0 + fabs(s)));
    }
    unsigned long long clk[4] = {0, 0, 0, 0};
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk));
    printf("%-28s M=%d N=%d K=%d  %8.1f us  %6.1f TF/s  maxrelerr %.2e  clk %.0f MHz\n", name, M, N, K, us, 2.0 * M * N * K / us / 1e6, maxerr,
           clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0);
    clk[0] = clk[1] = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), clk, sizeof clk));
    fflush(stdout);
    CK(hipMemset(C, 0, hC.size() * 4));
  };
#define RUN(WM, WN, KG, NBUF)                                                                                                              \
  {                                                                                                                                        \
    const int BM = 2 * WM * 32, BN = 2 * WN * 32;                                                                                          \
    if (M % BM == 0 && N % BN == 0 && K % KG == 0) {                                                                                       \
      const unsigned grid = (unsigned)((M / BM) * (N / BN));                                                                               \
      auto f = [&] { hipLaunchKernelGGL((gemm_direct<WM, WN, KG, NBUF>), dim3(grid), dim3(256), 0, 0, A, B, C, bias, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1); }; \
      check("direct " #WM "x" #WN " KG" #KG " NBUF" #NBUF, time_it(f, iters));                                                             \
    }                                                                                                                                      \
  }
  float* zeros;
  CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
#define RUND(WGM, WGN, WM, WN, BK, NS)                                                                                                      \
  {                                                                                                                                        \
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;                                                                                  \
    constexpr int ldsb = NS * (BM + BN) * BK * 4;                                                                                          \
    auto kern = gemm_dma<false, false, WGM, WGN, WM, WN, BK, NS>;                                                                          \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));                                          \
    const unsigned grid = (unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));                                                           \
    auto f = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(WGM * WGN * 64), ldsb, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1); }; \
    check("dma " #WGM "x" #WGN " w" #WM "x" #WN " BK" #BK " S" #NS, time_it(f, iters));                                                    \
  }
  {
    float* dst; CK(hipMalloc(&dst, 2048 * 256 * 4));
    for (int cfg = 0; cfg < 10; cfg++) {
      const int shape = cfg & 1 ? 16 : 32;
      const int nblk = cfg < 2 ? 2048 : (cfg < 4 ? 512 : (cfg < 6 ? 256 : (cfg < 8 ? 768 : 1024)));
      const int it = 4000 * (2048 / nblk);
      auto f = [&] { if (shape == 32) hipLaunchKernelGGL(mfma_only<32>, dim3(nblk), dim3(256), 0, 0, A, dst, it); else hipLaunchKernelGGL(mfma_only<16>, dim3(nblk), dim3(256), 0, 0, A, dst, it); };
      const float us = time_it(f, 3);
      const double flops = (double)nblk * 4 * it * (shape == 32 ? 32 * 4096.0 : 64 * 2048.0);
      printf("[%d blocks] ", nblk);
      unsigned long long clk[4]; CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk));
      printf("pure mfma %s: %.1f us  %.1f TF/s  clk %.0f MHz\n", shape == 32 ? "32x32x2" : "16x16x4", us, flops / us / 1e6, 100.0 * clk[0] / (double)clk[1]);
    }
  }
  {
    rocblas_handle h; rocblas_create_handle(&h);
    const float one = 1.0f, zero = 0.0f;
    auto f = [&] { rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, B, K, A, K, &zero, C, N); };
    const float us = time_it(f, iters);
    printf("rocblas_sgemm (no bias/relu)  %8.1f us  %6.1f TF/s\n", us, 2.0 * M * N * K / us / 1e6);
  }
#define RUNP(BK, WM, WN)                                                                                                                   \
  {                                                                                                                                        \
    constexpr int BM = 2 * WM * 32, BN = 2 * WN * 32;                                                                                      \
    constexpr int ldsb = 3 * (BM + BN) * BK * 4;                                                                                           \
    auto kern = gemm_plr<BK, WM, WN>;                                                                                                      \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));                                          \
    if (M % BM == 0 && N % BN == 0 && K % BK == 0) {                                                                                       \
      const unsigned grid = (unsigned)((M / BM) * (N / BN));                                                                               \
      auto f = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, A, B, C, bias, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1); }; \
      check("plr BK" #BK " w" #WM "x" #WN, time_it(f, iters));                                                                             \
    }                                                                                                                                      \
  }
  RUNP(16, 2, 2)
#define RUNPL(BK, WM, WN, LDSB)                                                                                                            \
  {                                                                                                                                        \
    constexpr int BM = 2 * WM * 32, BN = 2 * WN * 32;                                                                                      \
    auto kern = gemm_plr<BK, WM, WN>;                                                                                                      \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));                                          \
    const unsigned grid = (unsigned)((M / BM) * (N / BN));                                                                                 \
    auto f = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDSB, 0, A, B, C, bias, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1); }; \
    check("plr BK" #BK " w" #WM "x" #WN " lds" #LDSB, time_it(f, iters));                                                                  \
  }
#define RUNPA(BK, WM, WN, ABL)                                                                                                             \
  {                                                                                                                                        \
    constexpr int BM = 2 * WM * 32, BN = 2 * WN * 32;                                                                                      \
    constexpr int ldsb = 3 * (BM + BN) * BK * 4;                                                                                           \
    auto kern = gemm_plr<BK, WM, WN, ABL>;                                                                                                 \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));                                          \
    const unsigned grid = (unsigned)((M / BM) * (N / BN));                                                                                 \
    auto f = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, A, B, C, bias, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1); }; \
    const float us = time_it(f, iters);                                                                                                    \
    printf("plr BK" #BK " ablation " #ABL ": %.1f us  %.1f TF/s (results wrong by construction)\n", us, 2.0 * M * N * K / us / 1e6);      \
  }
  RUNPA(16, 2, 2, 1)
  RUNPA(16, 2, 2, 2)
  RUNPL(16, 2, 2, 72 * 1024)
  RUNPL(16, 2, 2, 120 * 1024)
  RUNP(32, 2, 2)
  RUNP(16, 4, 2)
  RUNP(16, 2, 4)
  RUNP(32, 4, 2)
  RUND(2, 2, 2, 2, 32, 2)
  RUND(4, 4, 2, 2, 16, 2)
  if (0) {
  RUND(2, 2, 2, 2, 32, 3)
  RUND(2, 2, 2, 2, 16, 3)
  RUND(2, 2, 2, 2, 16, 2)
  RUND(4, 2, 2, 2, 32, 2)
  RUND(2, 4, 2, 2, 32, 2)
  RUND(4, 2, 2, 2, 16, 3)
  RUND(2, 2, 2, 4, 32, 2)
  RUND(2, 2, 4, 2, 32, 2)
  RUND(4, 4, 2, 2, 16, 2)
  }
  if (0) {
  RUN(2, 2, 16, 2)
  RUN(2, 2, 16, 3)
  RUN(2, 2, 32, 2)
  RUN(2, 2, 32, 3)
  RUN(2, 4, 16, 2)
  RUN(2, 4, 16, 3)
  RUN(2, 4, 32, 2)
  RUN(4, 2, 16, 3)
  RUN(1, 2, 32, 3)
  RUN(1, 1, 32, 3)
  }
  return 0;
}
