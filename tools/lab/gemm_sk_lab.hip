// gemm_sk_lab.hip -- stand-alone lab for the round-3 fp32 GEMM main loop (not part of the product).
//
// One workgroup per CU (4 waves, one per SIMD), 128 x 128 x 64 tiles, v_mfma_f32_16x16x4_f32, and a software pipeline in which
// the matrix pipe never waits: per k-tile a wave issues 256 MFMAs back to back (8192 cycles) and every other instruction sits in
// the shadow of one of them --
//   * all 32 fragment reads of the k-tile (ds_read_b128: four k-steps of one 16-row tile, or -- rows-are-k operands -- one
//     k-step of four tiles) go out during the first 24 MFMAs, so the single LDS buffer is free early;
//   * barrier 1; the 16 ds_write_b128 of the NEXT k-tile (held in registers since the previous iteration) and the 16
//     buffer_load_dwordx4 of the one after it are spread one pair per 11 MFMAs;
//   * barrier 2; the first quarter of the next k-tile's fragments is read during the last 12 MFMAs.
// The k-tile stream is continuous across output tiles (persistent workgroups: the next tile's first operands are in flight
// while the previous tile's last MFMAs run), and the iteration space is split stream-K style for the weight-gradient form.
//   form 0  fwd: C[m][n] = act(sum_k A[m][k] B[n][k] + bias[n])      A, B k-contiguous
//   form 1  dX : C[m][n] = sum_k A[m][k] B[k][n]                      A k-contiguous, B rows-are-k
//   form 2  dW : C[m][n] += sum_k A[k][m] B[k][n]  (atomics, stream-K) A, B rows-are-k
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_sk_lab.hip -o tools/lab/gemm_sk_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <functional>
#include <type_traits>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_KC = 128 * 256 + 32 * 32;   // k-contiguous operand: row r at r*256 + (r>>2)*32 (32 B of padding per 1 KiB)
constexpr int LDS_KR = 64 * 512;              // rows-are-k operand: row k at k*512

struct SkArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int relu;
  int atomic;          // 1: C += partial sums by atomics, iteration space split evenly (stream-K)
  unsigned a_bytes, b_bytes;
  unsigned long long* clk;
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N_, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N_>{}, static_cast<F&&>(f)); }

#define PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x4 lds_read16(const char* base, int byte_off) {
  return *reinterpret_cast<const f32x4*>(base + byte_off);
}

template <bool AKR, bool BKR, bool ATOMIC, int DIAG = 0>   // DIAG (timing only, results wrong): 1 no global loads in the loop, 2 no LDS writes, 3 no barriers, 4 no fragment reads
__global__ __launch_bounds__(256, 1) void gemm_sk(const SkArgs g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int LDS_A = AKR ? LDS_KR : LDS_KC;
  char* const ldsA = lds;
  char* const ldsB = lds + LDS_A;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int c16 = lane & 15, q = lane >> 4;

  // fragment read addresses (bytes inside the operand's LDS image)
  const int fra = AKR ? (q * 2048 + wy * 256 + c16 * 16) : ((64 * wy + 4 * c16) * 256 + (16 * wy + c16) * 32 + q * 16);
  const int frb = BKR ? (q * 2048 + wx * 256 + c16 * 16) : ((64 * wx + 4 * c16) * 256 + (16 * wx + c16) * 32 + q * 16);
  // staging roles: where this thread's 8 + 8 float4 of a k-tile come from (byte offset inside the matrix, per-lane part) and go to
  const int srowA = AKR ? (tid >> 5) : (tid >> 4), schA = AKR ? (tid & 31) : (tid & 15);
  const int srowB = BKR ? (tid >> 5) : (tid >> 4), schB = BKR ? (tid & 31) : (tid & 15);
  const unsigned voffA = (unsigned)((srowA * g.lda + schA * 4) * 4);
  const unsigned voffB = (unsigned)((srowB * g.ldb + schB * 4) * 4);
  const int swA = AKR ? (srowA * 512 + schA * 16) : (srowA * 256 + (srowA >> 2) * 32 + schA * 16);
  const int swB = BKR ? (srowB * 512 + schB * 16) : (srowB * 256 + (srowB >> 2) * 32 + schB * 16);
  constexpr int SW_STEP_A = AKR ? 4096 : 4224, SW_STEP_B = BKR ? 4096 : 4224;     // LDS bytes between a thread's consecutive pieces
  const unsigned ioffA = (unsigned)((AKR ? 8 : 16) * g.lda * 4), ioffB = (unsigned)((BKR ? 8 : 16) * g.ldb * 4);   // global bytes between them
  const unsigned kadvA = AKR ? (unsigned)(BK * g.lda * 4) : (unsigned)(BK * 4), kadvB = BKR ? (unsigned)(BK * g.ldb * 4) : (unsigned)(BK * 4);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);

  // ---- this workgroup's share of the (tile, k-tile) iteration space ----
  const unsigned nbx = (unsigned)(g.N / BN), nby = (unsigned)(g.M / BM), ntiles = nbx * nby;
  const unsigned nk = (unsigned)(g.K / BK);
  const unsigned G = gridDim.x, w = blockIdx.x;
  // whole tiles: workgroup w takes tiles i*G + (w%8)*(G/8) + w/8, i = 0, 1, ... (blocks w, w+8, ... share an XCD: at any time an
  // XCD works on G/8 consecutive tiles, i.e. a few tile rows whose A panels stay in its L2)
  // stream-K (atomic): iterations [w*ipw, (w+1)*ipw) of the flat space, tile = it / nk
  const unsigned total_it = ntiles * nk;                  // < 2^32 for every layer of the path (host checks)
  const unsigned wperm = (w & 7u) * (G >> 3) + (w >> 3);
  unsigned it_b, it_e;
  if (ATOMIC) {
    const unsigned ipw = (total_it + G - 1) / G;
    it_b = w * ipw; it_e = it_b + ipw < total_it ? it_b + ipw : total_it;
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
    if (ATOMIC) lin = it_b / nk + c.seq;
    else lin = c.seq * G + wperm;
    if (lin >= ntiles) lin = ntiles - 1;                  // run-ahead loads past the end: any valid tile
    const unsigned by = lin / nbx, bx = lin - by * nbx;
    c.m0 = by * BM; c.n0 = bx * BN;
    c.offA = (AKR ? c.m0 * 4u : (unsigned)(c.m0 * g.lda * 4)) + c.kt * kadvA;
    c.offB = (BKR ? c.n0 * 4u : (unsigned)(c.n0 * g.ldb * 4)) + c.kt * kadvB;
  };
  auto advance = [&](Cursor& c) {
    c.kt++;
    if (c.kt == nk) { c.kt = 0; c.seq++; place(c); }
    else { c.offA += kadvA; c.offB += kadvB; }
  };
  Cursor ld{0, ATOMIC ? it_b % nk : 0u, 0, 0, 0, 0}, cp = ld;
  place(ld); place(cp);

  u32x4 P[16];
  auto gload_one = [&](int i, const Cursor& c) {
    if (i < 8) P[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, c.offA + (unsigned)i * ioffA, 0);
    else P[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, c.offB + (unsigned)(i - 8) * ioffB, 0);
  };
  auto lwrite_one = [&](int i) {
    if (i < 8) *reinterpret_cast<u32x4*>(ldsA + swA + i * SW_STEP_A) = P[i];
    else *reinterpret_cast<u32x4*>(ldsB + swB + (i - 8) * SW_STEP_B) = P[i];
  };
  f32x4 fa[4][4], fb[4][4];      // [j][r]: k-contiguous operand: r = 16-row tile, components = 4 k-steps; rows-are-k: r = k-step, components = 4 tiles
  auto fread = [&](int j, int r, bool isB) {
    if (!isB) fa[j][r] = lds_read16(ldsA, fra + (AKR ? (j * 8192 + r * 512) : (r * 256 + j * 64)));
    else fb[j][r] = lds_read16(ldsB, frb + (BKR ? (j * 8192 + r * 512) : (r * 256 + j * 64)));
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: k-tile 0 -> LDS, k-tile 1 -> registers, fragments j = 0 of k-tile 0 ----
#pragma unroll
  for (int i = 0; i < 16; i++) gload_one(i, ld);
  advance(ld);
#pragma unroll
  for (int i = 0; i < 16; i++) lwrite_one(i);
#pragma unroll
  for (int i = 0; i < 16; i++) gload_one(i, ld);
  advance(ld);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  PIN();
#pragma unroll
  for (int r = 0; r < 4; r++) { fread(0, r, false); fread(0, r, true); }
  PIN();

  unsigned long long c0 = 0, r0 = 0;
  if (g.clk && blockIdx.x == 0 && tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

  // outer loop: the output tiles (or stream-K segments) of this workgroup; inner loop: their k-tiles.  The operand stream
  // (cursor ld, two k-tiles ahead) does not know about the nest.
  for (unsigned it = 0; it < n_it;) {
    const unsigned seg = (nk - cp.kt) < (n_it - it) ? (nk - cp.kt) : (n_it - it);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 7" ::: "memory");
    for (unsigned kk = 0; kk < seg; kk++) {
      static_for<256>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int j = s >> 6, e = (s >> 4) & 3, tm = (s >> 2) & 3, tn = s & 3;
        const float av = AKR ? fa[j][e][tm] : fa[j][tm][e];
        const float bv = BKR ? fb[j][e][tn] : fb[j][tn][e];
        // accumulators pinned to AGPRs, destination tied to the addend: as a builtin hipcc gives the loop-carried accumulators
        // VGPR-class phis and shuffles all 64 through v_accvgpr_read / _write at the head of every iteration
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[tm][tn]) : "v"(av), "v"(bv));
        if constexpr (s < 24 && DIAG != 4) {       // fragments of k-groups 1..3
          constexpr int jj = 1 + s / 8, r = (s % 8) >> 1;
          fread(jj, r, (s & 1) != 0);
        }
        if constexpr (s == 53 && DIAG != 3) {      // every wave has its fragments: the LDS buffer may be overwritten
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (s >= 54 && s <= 219 && (s - 54) % 11 == 0) {
          constexpr int i = (s - 54) / 11;
          if constexpr (DIAG != 2) lwrite_one(i);  // k-tile it+1: registers -> LDS
          if constexpr (DIAG != 1) gload_one(i, ld);   // k-tile it+2 -> the same registers
        }
        if constexpr (s == 243 && DIAG != 3) {
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (s >= 244 && s < 252 && DIAG != 4) {   // first k-group of k-tile it+1
          constexpr int o = s - 244;               // order A0 B0 B1 B2 B3 A1 A2 A3
          if constexpr (o == 0) fread(0, 0, false);
          else if constexpr (o <= 4) fread(0, o - 1, true);
          else fread(0, o - 4, false);
        }
        PIN();
      });
      advance(ld);
    }
    it += seg;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // MFMA result -> v_accvgpr_read: the hazard recogniser does not see inline-asm MFMAs
    {
      const int nn = (int)cp.n0 + 64 * wx + 4 * c16;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!ATOMIC && g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + nn);
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int mm = (int)cp.m0 + 64 * wy + 16 * q + 4 * i + tm;
          f32x4 v = f32x4{acc[tm][0][i], acc[tm][1][i], acc[tm][2][i], acc[tm][3][i]};
          float* cp_ = g.C + (int64_t)mm * g.ldc + nn;
          if (ATOMIC) {
            atomicAdd(cp_ + 0, v.x); atomicAdd(cp_ + 1, v.y); atomicAdd(cp_ + 2, v.z); atomicAdd(cp_ + 3, v.w);
          } else {
            v += bv;
            const float lo = g.relu ? 0.f : -INFINITY;
            v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
            *reinterpret_cast<f32x4*>(cp_) = v;
          }
        }
    }
    cp.kt += seg;
    if (cp.kt == nk) { cp.kt = 0; cp.seq++; place(cp); }
  }
  if (g.clk && blockIdx.x == 0 && tid == 0) { g.clk[0] = __builtin_amdgcn_s_memtime() - c0; g.clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ------------------------------------------------------------------ host ------------------------------------------------------------------
static float time_it(const std::function<void()>& f, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; i++) f();     // the chip's clock takes tens of milliseconds of load to settle
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int Bt = argc > 1 ? atoi(argv[1]) : 32768;       // batch
  const int IN = argc > 2 ? atoi(argv[2]) : 1024, OUT = argc > 3 ? atoi(argv[3]) : 1024;
  const int relu_data = argc > 4 ? atoi(argv[4]) : 0;    // 1: x is half zeros (activations behind a ReLU)
  const int G = argc > 5 ? atoi(argv[5]) : 256;
  printf("layer %d -> %d at batch %d, %s operands, %d workgroups\n", IN, OUT, Bt, relu_data ? "post-ReLU x" : "dense random", G);
  std::vector<float> hx((size_t)Bt * IN), hw((size_t)OUT * IN), hb(OUT), hdy((size_t)Bt * OUT);
  uint64_t sd = 88172645463325252ull;
  auto rnd = [&] { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
  for (auto& v : hx) { v = rnd(); if (relu_data && v < 0) v = 0; }
  for (auto& v : hw) v = rnd() * 0.05f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hdy) v = rnd();
  float *x, *wt, *bias, *y, *dy, *dx, *dw; unsigned long long* clk;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&wt, hw.size() * 4)); CK(hipMalloc(&bias, OUT * 4)); CK(hipMalloc(&y, (size_t)Bt * OUT * 4));
  CK(hipMalloc(&dy, hdy.size() * 4)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&clk, 64));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), OUT * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
  auto k0 = gemm_sk<false, false, false>; auto k1 = gemm_sk<false, true, false>; auto k2 = gemm_sk<true, true, true>;
  CK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_KC));
  CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KC + LDS_KR));
  CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_KR));
  auto report = [&](const char* what, float us, double flops) {
    unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
    // matrix-pipe cycles the work needs per SIMD (256 CUs x 4 SIMDs, 32 cycles per 16x16x4 MFMA of 2048 MACs) over the cycles workgroup 0 ran
    const double need = flops / 2.0 / 2048.0 * 32.0 / 1024.0;
    printf("%-34s %9.1f us  %6.1f TF/s   in-kernel clock %4.0f MHz   matrix pipe busy %.3f of workgroup 0's cycles\n", what, us, flops / us / 1e6,
           c[1] ? 100.0 * c[0] / (double)c[1] : 0.0, c[0] ? need / (double)c[0] : 0.0);
  };
  const double fl = 2.0 * Bt * IN * OUT;
  // forward
  SkArgs a{}; a.A = x; a.B = wt; a.C = y; a.bias = bias; a.M = Bt; a.N = OUT; a.K = IN; a.lda = IN; a.ldb = IN; a.ldc = OUT; a.relu = 1; a.atomic = 0;
  a.a_bytes = (unsigned)(hx.size() * 4); a.b_bytes = (unsigned)(hw.size() * 4); a.clk = clk;
  report("fwd  (kc,kc)", time_it([&] { hipLaunchKernelGGL(k0, dim3(G), dim3(256), 2 * LDS_KC, 0, a); }, 30), fl);
  if (argc > 6 && atoi(argv[6])) {
#define DIAGRUN(D, NAME) { auto kd = gemm_sk<false, false, false, D>; CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_KC)); \
      report("fwd  " NAME, time_it([&] { hipLaunchKernelGGL(kd, dim3(G), dim3(256), 2 * LDS_KC, 0, a); }, 30), fl); }
    DIAGRUN(1, "no global loads (wrong)")
    DIAGRUN(2, "no LDS writes (wrong)")
    DIAGRUN(3, "no barriers (wrong)")
    DIAGRUN(4, "no fragment reads (wrong)")
    report("fwd  (kc,kc) again", time_it([&] { hipLaunchKernelGGL(k0, dim3(G), dim3(256), 2 * LDS_KC, 0, a); }, 30), fl);
  }
  {
    std::vector<float> hy((size_t)Bt * OUT); CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4000; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % Bt), n = (int)(((uint64_t)t * 40503u + 7) % OUT);
      double s = hb[n], mass = fabs(hb[n]);
      for (int k = 0; k < IN; k++) { s += (double)hx[(size_t)m * IN + k] * hw[(size_t)n * IN + k]; mass += fabs((double)hx[(size_t)m * IN + k] * hw[(size_t)n * IN + k]); }
      if (s < 0) s = 0;
      worst = fmax(worst, fabs(hy[(size_t)m * OUT + n] - s) / (mass + 1e-30));
    }
    printf("   fwd check: worst |err| / term mass over 4000 samples = %.3e %s\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  }
  // dX = dy W
  SkArgs b{}; b.A = dy; b.B = wt; b.C = dx; b.M = Bt; b.N = IN; b.K = OUT; b.lda = OUT; b.ldb = IN; b.ldc = IN; b.relu = 0; b.atomic = 0;
  b.a_bytes = (unsigned)(hdy.size() * 4); b.b_bytes = (unsigned)(hw.size() * 4); b.clk = clk;
  report("dX   (kc,kr)", time_it([&] { hipLaunchKernelGGL(k1, dim3(G), dim3(256), LDS_KC + LDS_KR, 0, b); }, 30), fl);
  {
    std::vector<float> hd(hx.size()); CK(hipMemcpy(hd.data(), dx, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4000; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % Bt), n = (int)(((uint64_t)t * 40503u + 7) % IN);
      double s = 0, mass = 0;
      for (int k = 0; k < OUT; k++) { s += (double)hdy[(size_t)m * OUT + k] * hw[(size_t)k * IN + n]; mass += fabs((double)hdy[(size_t)m * OUT + k] * hw[(size_t)k * IN + n]); }
      worst = fmax(worst, fabs(hd[(size_t)m * IN + n] - s) / (mass + 1e-30));
    }
    printf("   dX  check: worst |err| / term mass over 4000 samples = %.3e %s\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  }
  // dW += dy^T x   (stream-K, atomics)
  SkArgs c{}; c.A = dy; c.B = x; c.C = dw; c.M = OUT; c.N = IN; c.K = Bt; c.lda = OUT; c.ldb = IN; c.ldc = IN; c.relu = 0; c.atomic = 1;
  c.a_bytes = (unsigned)(hdy.size() * 4); c.b_bytes = (unsigned)(hx.size() * 4); c.clk = clk;
  CK(hipMemset(dw, 0, hw.size() * 4));
  hipLaunchKernelGGL(k2, dim3(G), dim3(256), 2 * LDS_KR, 0, c);
  {
    std::vector<float> hd(hw.size()); CK(hipMemcpy(hd.data(), dw, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 300; t++) {
      const int m = (int)(((uint64_t)t * 2654435761u) % OUT), n = (int)(((uint64_t)t * 40503u + 7) % IN);
      double s = 0, mass = 0;
      for (int k = 0; k < Bt; k++) { s += (double)hdy[(size_t)k * OUT + m] * hx[(size_t)k * IN + n]; mass += fabs((double)hdy[(size_t)k * OUT + m] * hx[(size_t)k * IN + n]); }
      worst = fmax(worst, fabs(hd[(size_t)m * IN + n] - s) / (mass + 1e-30));
    }
    printf("   dW  check: worst |err| / term mass over 300 samples = %.3e %s\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  }
  report("dW   (kr,kr) stream-K", time_it([&] { hipLaunchKernelGGL(k2, dim3(G), dim3(256), 2 * LDS_KR, 0, c); }, 30), fl);
  return 0;
}
