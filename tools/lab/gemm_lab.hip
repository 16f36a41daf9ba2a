// gemm_lab.hip -- stand-alone bench for experimental fp32 MFMA GEMM kernels (not part of the product).
//   C[m][n] = act(sum_k A[m][k] * B[n][k] + bias[n])   both operands k-contiguous (the Linear forward form)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_lab.hip -o tools/lab/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ unsigned long long g_ts[64 * 8];

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM x BN block tile, BK = 64, 4 waves; each wave owns a 32x32 sub-tile over 1/KS of every k-tile.
template <int BM, int BN, int NSTAGE, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_kk_glds(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    const float* __restrict__ bias, const float* __restrict__ zeros, int M, int N, int K,
                                                    int64_t lda, int64_t ldb, int64_t ldc, int relu) {
  constexpr int KS = (64 / BM) * (64 / BN);
  constexpr int A_CH = BM * 16, B_CH = BN * 16, STAGE_CH = A_CH + B_CH;
  constexpr int NI = STAGE_CH / 64, NIW = NI / 4;       // DMA wave-instructions per stage, per wave
  constexpr int JW = 8 / KS;                            // k-octets of a k-tile per wave
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of tiles
  int bx, by;
  {
    const unsigned nbx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.y * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3, q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx); by = (int)(nlin / nbx);
  }
  const int m0 = by * BM, n0 = bx * BN;
  int wm, wn, ks;
  if (KS == 1) { wm = wave >> 1; wn = wave & 1; ks = 0; }
  else if (KS == 2 && BM == 32) { wm = 0; wn = wave & 1; ks = wave >> 1; }
  else if (KS == 2) { wm = wave & 1; wn = 0; ks = wave >> 1; }
  else { wm = 0; wn = 0; ks = wave; }

  // per-lane source rows of this wave's DMA instructions
  const float* src_row[NIW];
  int src_c4[NIW];
#pragma unroll
  for (int i = 0; i < NIW; i++) {
    const int q = wave + 4 * i;
    const int ch = q * 64 + lane;
    if (ch < A_CH) {
      const int r = ch >> 4, s = ch & 15, c = s ^ (r & 15);
      const int gm = m0 + r;
      src_row[i] = gm < M ? A + (int64_t)gm * lda + 4 * c : nullptr;
      src_c4[i] = 4 * c;
    } else {
      const int cb = ch - A_CH;
      const int r = cb >> 4, s = cb & 15, c = s ^ (r & 15);
      const int gn = n0 + r;
      src_row[i] = gn < N ? B + (int64_t)gn * ldb + 4 * c : nullptr;
      src_c4[i] = 4 * c;
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
  auto issue = [&](int kt, int buf, int i0, int i1) {
    const int k0 = kt * 64;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      const int q = wave + 4 * i;
      const float* src = (src_row[i] != nullptr && k0 + src_c4[i] < K) ? src_row[i] + k0 : zeros;
      // inline asm: hipcc's waitcnt pass must not see the LDS-DMA, or it drains vmcnt(0) before every ds_read
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE_CH + q * 64) * 16u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;

  const int nk = (K + 63) / 64;
  if (MODE != 2 && MODE != 3) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; s++) issue(s, s, 0, NIW);
  }

  const int ra = wm * 32 + lr, rb = wn * 32 + lr;
  int buf = 0, nbuf = NSTAGE - 1;
  for (int t = 0; t < nk; t++) {
    const bool ts_on = MODE == 4 && blockIdx.x == 3 && blockIdx.y == 5 && wave == 1 && t < 8;
    if (ts_on && lane == 0) g_ts[t * 8 + 0] = __builtin_amdgcn_s_memtime();
    if (MODE != 2 && MODE != 3) wait_vmcnt<(NSTAGE - 2) * NIW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ts_on && lane == 0) g_ts[t * 8 + 1] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if (ts_on && lane == 0) g_ts[t * 8 + 2] = __builtin_amdgcn_s_memtime();
    if (MODE != 2 && MODE != 3 && MODE != 5) issue(t + NSTAGE - 1, nbuf, 0, NIW);
    if (ts_on && lane == 0) g_ts[t * 8 + 3] = __builtin_amdgcn_s_memtime();
    const float4* as = lds + buf * STAGE_CH;
    const float4* bs = as + A_CH;
    auto rd = [&](int j, float4& a, float4& b) {
      const int c = 2 * (ks * JW + j) + lh;
      if (MODE == 3) { a = make_float4(t, j, lane, 1.f); b = make_float4(j, t, 2.f, lane); return; }
      a = as[ra * 16 + (c ^ (ra & 15))];
      b = bs[rb * 16 + (c ^ (rb & 15))];
    };
    float4 a_cur, b_cur, a_nxt, b_nxt;
    rd(0, a_cur, b_cur);
#pragma unroll
    for (int j = 0; j < JW; j++) {
      if (j + 1 < JW) rd(j + 1, a_nxt, b_nxt);
      if (MODE == 5) issue(t + NSTAGE - 1, nbuf, j * NIW / JW, (j + 1) * NIW / JW);   // DMA issue spread over the MFMA chain
      if (MODE == 1) {
        acc[0] += a_cur.x * b_cur.x + a_cur.y * b_cur.y + a_cur.z * b_cur.z + a_cur.w * b_cur.w;
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc, 0, 0, 0);
      }
      a_cur = a_nxt; b_cur = b_nxt;
    }
    if (ts_on && lane == 0) g_ts[t * 8 + 4] = __builtin_amdgcn_s_memtime();
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }
  wait_vmcnt<0>();
  if (KS > 1) {
    // meet the k-slices in LDS: ((s0 + s1) + s2) + s3
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (ks != 0) return;
    // group of KS waves with the same (wm, wn): wave ids wave + g * (4 / KS)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = acc[r];
#pragma unroll
      for (int g = 1; g < KS; g++) v += red[((wave + g * (4 / KS)) * 16 + r) * 64 + lane];
      acc[r] = v;
    }
  }
  const int n = n0 + wn * 32 + lr;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) {
        float v = acc[r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
  }
}


// 64x64 block tile, BK = 64, NW waves = 4 (2x2 sub-tiles of 32x32) x KS k-slices of every k-tile (intra-block split-K):
// more waves per SIMD on the same tile, so that one wave's LDS-DMA issue stalls sit under another wave's MFMAs.
template <int NW, int NSTAGE>
__global__ __launch_bounds__(NW * 64) void gemm_kk64(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    const float* __restrict__ bias, const float* __restrict__ zeros, int M, int N, int K,
                                                    int64_t lda, int64_t ldb, int64_t ldc, int relu) {
  constexpr int KS = NW / 4;
  constexpr int A_CH = 64 * 16, STAGE_CH = 2 * A_CH;
  constexpr int NIW = (STAGE_CH / 64) / NW;             // DMA pieces per stage per wave
  constexpr int JW = 8 / KS;                            // k-octets of a k-tile per wave
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by;
  {
    const unsigned nbx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.y * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3, q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx); by = (int)(nlin / nbx);
  }
  const int m0 = by * 64, n0 = bx * 64;
  const int sub = wave & 3, ks = wave >> 2;
  const int wm = sub >> 1, wn = sub & 1;

  const float* src_row[NIW];
  int src_c4[NIW];
#pragma unroll
  for (int i = 0; i < NIW; i++) {
    const int q = wave + NW * i;
    const int ch = q * 64 + lane;
    const bool isA = ch < A_CH;
    const int cb = isA ? ch : ch - A_CH;
    const int r = cb >> 4, s = cb & 15, c = s ^ (r & 15);
    const int g = (isA ? m0 : n0) + r;
    const bool ok = g < (isA ? M : N);
    src_row[i] = ok ? (isA ? A + (int64_t)g * lda : B + (int64_t)g * ldb) + 4 * c : nullptr;
    src_c4[i] = 4 * c;
  }
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
  auto issue = [&](int kt, int buf, int i0, int i1) {
    const int k0 = kt * 64;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      const int q = wave + NW * i;
      const float* src = (src_row[i] != nullptr && k0 + src_c4[i] < K) ? src_row[i] + k0 : zeros;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE_CH + q * 64) * 16u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  const int nk = (K + 63) / 64;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) issue(s, s, 0, NIW);
  const int ra = wm * 32 + lr, rb = wn * 32 + lr;
  int buf = 0, nbuf = NSTAGE - 1;
  for (int t = 0; t < nk; t++) {
    wait_vmcnt<(NSTAGE - 2) * NIW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* as = lds + buf * STAGE_CH;
    const float4* bs = as + A_CH;
    auto rd = [&](int j, float4& a, float4& b) {
      const int c = 2 * (ks * JW + j) + lh;
      a = as[ra * 16 + (c ^ (ra & 15))];
      b = bs[rb * 16 + (c ^ (rb & 15))];
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
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }
  wait_vmcnt<0>();
  constexpr int RPW = 16 / KS;                          // accumulator registers finished by each wave
  if (KS > 1) {
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(lds);       // [ks][sub][16][64]
#pragma unroll
    for (int r = 0; r < 16; r++) red[((ks * 4 + sub) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPW; q++) {
      const int r = ks * RPW + q;
      float v = red[((0 * 4 + sub) * 16 + r) * 64 + lane];
#pragma unroll
      for (int g = 1; g < KS; g++) v += red[((g * 4 + sub) * 16 + r) * 64 + lane];
      acc[q] = v;
    }
  }
  const int n = n0 + wn * 32 + lr;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int q = 0; q < RPW; q++) {
      const int r = KS > 1 ? ks * RPW + q : q;
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) {
        float v = acc[q] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
  }
}

template <int NW, int NSTAGE>
float run64(const float* A, const float* B, float* C, const float* bias, const float* zeros, int M, int N, int K, int iters) {
  constexpr int lds_bytes = NSTAGE * 128 * 16 * 16;
  auto kern = gemm_kk64<NW, NSTAGE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}


// 128x128 block tile, BK = 32, 16 waves as 4x4 sub-tiles of 32x32 (no k-split), NSTAGE stages of 32 KB.
// LDS image per operand: [128 rows][8 chunks of 16 B], slot = chunk ^ ((row >> 1) & 7)  (conflict-free ds_read_b128)
template <int NSTAGE>
__global__ __launch_bounds__(1024) void gemm_kk128(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   const float* __restrict__ bias, const float* __restrict__ zeros, int M, int N, int K,
                                                   int64_t lda, int64_t ldb, int64_t ldc, int relu) {
  constexpr int NW = 16;
  constexpr int A_CH = 128 * 8, STAGE_CH = 2 * A_CH;    // 2048 chunks = 32 KB
  constexpr int NIW = (STAGE_CH / 64) / NW;             // 2 DMA pieces per stage per wave
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int bx, by;
  {
    const unsigned nbx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.y * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, loc = lin >> 3, q = total >> 3, rem = total & 7u;
    const unsigned nlin = xcd * q + (xcd < rem ? xcd : rem) + loc;
    bx = (int)(nlin % nbx); by = (int)(nlin / nbx);
  }
  const int m0 = by * 128, n0 = bx * 128;
  const int wm = wave >> 2, wn = wave & 3;
  const float* src_row[NIW];
  int src_c4[NIW];
#pragma unroll
  for (int i = 0; i < NIW; i++) {
    const int q = wave + NW * i;
    const int ch = q * 64 + lane;
    const bool isA = ch < A_CH;
    const int cb = isA ? ch : ch - A_CH;
    const int r = cb >> 3, s = cb & 7, c = s ^ ((r >> 1) & 7);
    const int g = (isA ? m0 : n0) + r;
    const bool ok = g < (isA ? M : N);
    src_row[i] = ok ? (isA ? A + (int64_t)g * lda : B + (int64_t)g * ldb) + 4 * c : nullptr;
    src_c4[i] = 4 * c;
  }
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)lds;
  auto issue = [&](int kt, int buf, int i0, int i1) {
    const int k0 = kt * 32;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      const int q = wave + NW * i;
      const float* src = (src_row[i] != nullptr && k0 + src_c4[i] < K) ? src_row[i] + k0 : zeros;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE_CH + q * 64) * 16u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  const int nk = (K + 31) / 32;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) issue(s, s, 0, NIW);
  const int ra = wm * 32 + lr, rb = wn * 32 + lr;
  int buf = 0, nbuf = NSTAGE - 1;
  for (int t = 0; t < nk; t++) {
    wait_vmcnt<(NSTAGE - 2) * NIW>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4* as = lds + buf * STAGE_CH;
    const float4* bs = as + A_CH;
    auto rd = [&](int j, float4& a, float4& b) {
      const int c = 2 * j + lh;
      a = as[ra * 8 + (c ^ ((ra >> 1) & 7))];
      b = bs[rb * 8 + (c ^ ((rb >> 1) & 7))];
    };
    float4 a_cur, b_cur, a_nxt, b_nxt;
    rd(0, a_cur, b_cur);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j + 1 < 4) rd(j + 1, a_nxt, b_nxt);
      if (j < NIW) issue(t + NSTAGE - 1, nbuf, j, j + 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc, 0, 0, 0);
      a_cur = a_nxt; b_cur = b_nxt;
    }
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }
  wait_vmcnt<0>();
  const int n = n0 + wn * 32 + lr;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) {
        float v = acc[r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        C[(int64_t)m * ldc + n] = v;
      }
    }
  }
}

template <int NSTAGE>
float run128(const float* A, const float* B, float* C, const float* bias, const float* zeros, int M, int N, int K, int iters) {
  constexpr int lds_bytes = NSTAGE * 2048 * 16;
  auto kern = gemm_kk128<NSTAGE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  dim3 grid((N + 127) / 128, (M + 127) / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

template <int BM, int BN, int NSTAGE, int MODE = 0>
float run(const float* A, const float* B, float* C, const float* bias, const float* zeros, int M, int N, int K, int iters) {
  constexpr int lds_bytes = NSTAGE * (BM + BN) * 16 * 16;
  auto kern = gemm_kk_glds<BM, BN, NSTAGE, MODE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, 0, A, B, C, bias, zeros, M, N, K, (int64_t)K, (int64_t)K, (int64_t)N, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 432;
  const int iters = (double)M * N * K > 4e9 ? 30 : 200;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N), hC((size_t)M * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.0f;
  float *A, *B, *C, *bias, *zeros;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, hC.size() * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&zeros, 256));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(zeros, 0, 256));
  auto check = [&](const char* name, float us) {
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 4000; t++) {
      const int m = (t * 7919) % M, n = (t * 104729) % N;
      double s = hb[n];
      for (int k = 0; k < K; k++) s += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
      if (s < 0) s = 0;
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)m * N + n]) / (1.0 + fabs(s)));
    }
    printf("%-22s M=%d N=%d K=%d  %7.2f us  %6.1f TF/s  maxrelerr %.2e\n", name, M, N, K, us, 2.0 * M * N * K / us / 1e6, maxerr);
    CK(hipMemset(C, 0, hC.size() * 4));
  };
  check("kk128 16w s3", run128<3>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk128 16w s4", run128<4>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk64 4w s4", run64<4, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk64 8w s4", run64<8, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk64 8w s3", run64<8, 3>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk64 16w s4", run64<16, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("kk64 16w s3", run64<16, 3>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s4", run<64, 64, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s4 noMFMA", run<64, 64, 4, 1>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s4 noGLDS", run<64, 64, 4, 2>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s4 MFMAonly", run<64, 64, 4, 3>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s4 noMFMA", run<32, 32, 4, 1>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s4 noGLDS", run<32, 32, 4, 2>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s4 MFMAonly", run<32, 32, 4, 3>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s4 timestamps", run<64, 64, 4, 4>(A, B, C, bias, zeros, M, N, K, 1));
  {
    unsigned long long ts[64];
    CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
    for (int t = 0; t < 7; t++)
      printf("  stage %d: +%6llu | vmcnt wait %5llu  barrier %5llu  issue %5llu  reads+mfma %5llu\n", t, ts[t * 8] - ts[0], ts[t * 8 + 1] - ts[t * 8],
             ts[t * 8 + 2] - ts[t * 8 + 1], ts[t * 8 + 3] - ts[t * 8 + 2], ts[t * 8 + 4] - ts[t * 8 + 3]);
  }
  check("64x64 s4 spread", run<64, 64, 4, 5>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s3 spread", run<64, 64, 3, 5>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x64 s4 spread", run<32, 64, 4, 5>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s4 spread", run<32, 32, 4, 5>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x64 s3", run<64, 64, 3>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x64 s4", run<32, 64, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("64x32 s4", run<64, 32, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s4", run<32, 32, 4>(A, B, C, bias, zeros, M, N, K, iters));
  check("32x32 s3", run<32, 32, 3>(A, B, C, bias, zeros, M, N, K, iters));
  return 0;
}
