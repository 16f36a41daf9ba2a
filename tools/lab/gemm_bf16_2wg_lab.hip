// gemm_bf16_2wg_lab.hip -- stand-alone lab for the tensor-op (bf16 operands, fp32 accumulate) GEMM with TWO independent workgroups per CU
// (not part of the product).  OUTCOME (round 6): correct, and not taken -- with the DMA removed from its loop this kernel runs at 1.5 PFLOP/s, with the
// MFMAs removed it takes as long as with them: it is bound by the rate LDS-DMA fills LDS (~11 TB/s chip-wide on either tile shape), and a 128 x 256
// tile needs 1.5x the operand bytes per flop of the 256 x 256 one.  Wired into the library for the forward of layers up to 1024 deep it was level
// alone (32768 x 1024 -> 1024: 101 vs 96 us through the C-ABI) and slower in the step (120 vs 99 us).  profiles/r06_lab_gemm_bf16_2wg.txt.
//
// Why: csrc/linear_bf16_dma.hip runs one workgroup of 8 waves per CU on a 256 x 256 tile with all 160 KB of LDS.  Its main loop reaches
// ~1.4 PFLOP/s, but every workgroup of a round reaches its epilogue at the same time: 256 tiles x 256 x 256 x 6 bytes (fp32 + bf16 twin) leave
// in one burst that HBM takes ~20 us to absorb while no MFMA runs anywhere -- 40 of the 96 us of 32768 x 1024 -> 1024 forward.  Here a workgroup
// is 4 waves on a 128 x 256 tile with 72 KB of LDS, two of them share a CU (one wave of each per SIMD) and nothing synchronises them: one's
// epilogue runs under the other's MFMAs, and the workgroups of a launch drift apart.
//
//   * 4 waves = 2 (row groups of 64) x 2 (column groups of 128); a wave owns 64 x 128: 4 x 8 accumulators of 16 x 16.
//   * a k-step is 32 deep: A unit 128 rows x 64 B = 8 KB, B-lo / B-hi units 128 columns each; a wave reads the A unit (its 4 fragments) and
//     ONE B unit (8 fragments): 12 ds_read_b128 for 32 MFMAs.  Ring of three stages x 24 KB, filled by LDS-DMA two stages ahead:
//         { s_waitcnt vmcnt(6);  s_barrier;  6 DMA pieces of stage t + 2;  12 fragment reads of stage t;  32 MFMAs }
//   * LDS images lane-linear; k-contiguous unit: chunk j (16 B) of row u at slot j ^ ((u >> 2) & 3); rows-are-k unit (32 k-rows x 256 B):
//     chunk j (8 columns) of k-row r at slot j ^ (((r & 3) << 2) | ((r >> 2) & 3)), read with ds_read_b64_tr_b16.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_bf16_2wg_lab.hip -o tools/lab/gemm_bf16_2wg_lab
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

constexpr int T_BM = 128, T_BN = 256, T_BK = 32;
constexpr int T_UNIT = 8192, T_STAGE = 3 * T_UNIT, T_NS = 3;
constexpr int T_LDS = T_NS * T_STAGE;             // 72 KB: two workgroups per CU; the epilogue reuses it (18 KB per wave)
enum { T_EPI_FWD = 0, T_EPI_DX = 1 };

struct TArgs {
  const unsigned short* A; const unsigned short* B;
  float* C; unsigned short* C16;
  const float* bias;
  const float* mask; const unsigned short* mask16;
  int64_t lda, ldb, ldc, ldmask;
  int M, N, K;
  int act;            // 1: relu
  int add;
  unsigned a_bytes, b_bytes;
};

#define T_FENCE() __builtin_amdgcn_sched_barrier(0)
#define T_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define T_BARRIER() __builtin_amdgcn_s_barrier()

__device__ __forceinline__ void t_glds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned dst, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

// DIAG: 0 product; 1 no DMA in the loop; 2 no fragment reads; 3 no MFMAs; 4 no epilogue
template <bool BKR, int EPI, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_2wg_kernel(const TArgs g) {
  extern __shared__ __attribute__((aligned(16))) char t_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c = lane & 15, q = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // tile of this workgroup: n fastest, the workgroups of one XCD take neighbouring tiles
  const unsigned nbx = (unsigned)((g.N + T_BN - 1) / T_BN), nby = (unsigned)((g.M + T_BM - 1) / T_BM), ntiles = nbx * nby;
  const unsigned total = gridDim.x, w = blockIdx.x;
  const unsigned xcd = w & 7u, loc = w >> 3, qq = total >> 3, rem = total & 7u;
  const unsigned tile = xcd * qq + (xcd < rem ? xcd : rem) + loc;
  if (tile >= ntiles) return;
  const unsigned by = tile / nbx, bx = tile - by * nbx;
  const int m0 = (int)by * T_BM, n0 = (int)bx * T_BN;
  const int nk = g.K / T_BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.B), 0, g.b_bytes, 0x00020000);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)t_lds;

  // staging: per-lane part of the source offset; a round of the 256 threads writes 4 KB = 64 unit-rows (k-contiguous) or 16 k-rows (rows-are-k)
  unsigned voffA, voffB;
  {
    const int u = tid >> 2, j = (tid & 3) ^ ((u >> 2) & 3);
    const int kr = tid >> 4, jr = (tid & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
    voffA = (unsigned)((u * g.lda + j * 8) * 2);
    voffB = BKR ? (unsigned)((kr * g.ldb + jr * 8) * 2) : (unsigned)((u * g.ldb + j * 8) * 2);
  }
  const unsigned istepA = (unsigned)(64 * g.lda * 2), istepB = (unsigned)((BKR ? 16 : 64) * g.ldb * 2);
  auto stage = [&](const int t, const int ring) {
    const unsigned slot = lds_base + (unsigned)(ring * T_STAGE) + (unsigned)wave * 1024u;
    const bool live = t < nk;
    {
      const unsigned soff = live ? (unsigned)(((int64_t)m0 * g.lda + (int64_t)t * T_BK) * 2) : g.a_bytes;
      t_glds16(voffA, rsA, slot, soff);
      t_glds16(voffA, rsA, slot + 4096u, live ? soff + istepA : g.a_bytes);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int o0 = n0 + h * 128;
      const unsigned soff = !live ? g.b_bytes : BKR ? (unsigned)(((int64_t)t * T_BK * g.ldb + o0) * 2) : (unsigned)(((int64_t)o0 * g.ldb + (int64_t)t * T_BK) * 2);
      t_glds16(voffB, rsB, slot + (unsigned)((1 + h) * T_UNIT), soff);
      t_glds16(voffB, rsB, slot + (unsigned)((1 + h) * T_UNIT) + 4096u, live ? soff + istepB : g.b_bytes);
    }
  };

  // fragment read offsets inside a unit
  const int kcA = (wr * 64 + c) * 64 + ((q ^ (c >> 2)) << 4);          // + f * 1024
  const int kcB = c * 64 + ((q ^ (c >> 2)) << 4);                      // + f * 1024, f = 0..7 (the wave's B unit)
  const int krX1 = (tq << 2) | (2 * (q & 1)), krX2 = krX1 | 1;
  const int krRow = (8 * q + tq) * 256 + 8 * (tp & 1);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
  auto fragB = [&](const char* unit, int f) -> bf16x8 {
    if (!BKR) return *reinterpret_cast<const bf16x8*>(unit + kcB + f * 1024);
    const int ch = ((f * 16) >> 3) + (tp >> 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(unit + krRow + ((ch ^ krX1) << 4)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(unit + krRow + 1024 + ((ch ^ krX2) << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0); stage(1, 1);
  int rs = 0, ws = 2;          // ring slot read in this step / written in this step
  for (int t = 0; t < nk; t++) {
    T_WAIT_VM(6);
    T_BARRIER();
    T_FENCE();
    if (DIAG != 1 || t == 0) stage(t + 2, ws);
    ws = ws == T_NS - 1 ? 0 : ws + 1;
    T_FENCE();
    const char* sb = t_lds + rs * T_STAGE;
    rs = rs == T_NS - 1 ? 0 : rs + 1;
    const char* ub = sb + (1 + wc) * T_UNIT;
    bf16x8 a[4], b[8];
    if (DIAG != 2) {
      b[0] = fragB(ub, 0);
#pragma unroll
      for (int f = 0; f < 4; f++) a[f] = *reinterpret_cast<const bf16x8*>(sb + kcA + f * 1024);
#pragma unroll
      for (int f = 1; f < 8; f++) b[f] = fragB(ub, f);
    } else {
#pragma unroll
      for (int f = 0; f < 4; f++) a[f] = __builtin_bit_cast(bf16x8, s16x8{(short)t, 1, 2, 3, 4, 5, 6, (short)lane});
#pragma unroll
      for (int f = 0; f < 8; f++) b[f] = __builtin_bit_cast(bf16x8, s16x8{(short)f, 1, 2, 3, 4, 5, 6, (short)t});
    }
    if (DIAG != 3) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tn = 0; tn < 8; tn++)
#pragma unroll
        for (int tm = 0; tm < 4; tm++)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tn], a[tm], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int tn = 0; tn < 8; tn++)
#pragma unroll
        for (int tm = 0; tm < 4; tm++) acc[tm][tn][0] += (float)a[tm][0] + (float)b[tn][1];
    }
    T_FENCE();
  }
  T_WAIT_VM(0);
  T_BARRIER();
  if (DIAG == 4) {
    float s = 0.f;
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 8; tn++) s += acc[tm][tn][0] + acc[tm][tn][1] + acc[tm][tn][2] + acc[tm][tn][3];
    if (s == 12345.678f) g.C[0] = s;
    return;
  }

  // epilogue: lane (c, q) holds C[tm * 16 + c][tn * 16 + 4 q + {0..3}] of the wave's 64 x 128; half of it (64 columns) at a time through the
  // wave's own 18 KB: rows padded to 272 B; back out as 8 rows x 128 contiguous bytes per store instruction
  constexpr int EP_LD = 272;
  char* blk = t_lds + wave * 18432;
  const int rowb = m0 + wr * 64;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int colb = n0 + wc * 128 + half * 64;
    f32x4 bias_r[4];
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
      bias_r[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == T_EPI_FWD) {
        const int cc = colb + tn * 16 + 4 * q;
        if (g.bias && cc < g.N) bias_r[tn] = *reinterpret_cast<const f32x4*>(g.bias + cc);
      }
    }
    const int rr = lane >> 3, rc = lane & 7;
    f32x4 mk[16]; s16x4 mh[16];
    if constexpr (EPI == T_EPI_DX) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int p = e >> 1, h = e & 1;
        const int row = rowb + p * 8 + rr, col = colb + h * 32 + rc * 4;
        const bool in = row < g.M && col < g.N;
        mk[e] = f32x4{0.f, 0.f, 0.f, 0.f}; mh[e] = s16x4{0, 0, 0, 0};
        if (g.mask16) { if (in) mh[e] = *reinterpret_cast<const s16x4*>(g.mask16 + (int64_t)row * g.ldmask + col); }
        else if (g.mask) { if (in) mk[e] = *reinterpret_cast<const f32x4*>(g.mask + (int64_t)row * g.ldmask + col); }
      }
    }
#pragma unroll
    for (int tm = 0; tm < 4; tm++)
#pragma unroll
      for (int tn = 0; tn < 4; tn++) {
        f32x4 v = acc[tm][4 * half + tn];
        if constexpr (EPI == T_EPI_FWD) {
          v += bias_r[tn];
          if (g.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        *reinterpret_cast<f32x4*>(blk + (tm * 16 + c) * EP_LD + (tn * 16 + 4 * q) * 4) = v;
      }
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        f32x4 v = *reinterpret_cast<const f32x4*>(blk + (p * 8 + rr) * EP_LD + (h * 32 + rc * 4) * 4);
        const int row = rowb + p * 8 + rr, col = colb + h * 32 + rc * 4;
        if (row < g.M && col < g.N) {
          float* cp = g.C + (int64_t)row * g.ldc + col;
          if constexpr (EPI == T_EPI_DX) {
            if (g.mask16) {
              const s16x4 m = mh[p * 2 + h];
              v.x = m[0] > 0 ? v.x : 0.f; v.y = m[1] > 0 ? v.y : 0.f; v.z = m[2] > 0 ? v.z : 0.f; v.w = m[3] > 0 ? v.w : 0.f;
            } else if (g.mask) {
              const f32x4 m = mk[p * 2 + h];
              v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
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

// ------------------------------------------------------------------------------------------------------------------------------
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static float time_it(const std::function<void()>& f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
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
template <typename K> static void set_lds(K k) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS)); }

int main(int argc, char** argv) {
  const int Bt = argc > 1 ? atoi(argv[1]) : 32768;
  const int IN = argc > 2 ? atoi(argv[2]) : 1024;
  const int OUT = argc > 3 ? atoi(argv[3]) : 1024;
  const int diag = argc > 4 ? atoi(argv[4]) : 0;
  const int iters = 30;
  printf("layer %d -> %d at batch %d, bf16 operands from twins, fp32 accumulate; 128 x 256 tiles, two workgroups per CU\n", IN, OUT, Bt);
  std::vector<uint16_t> hx((size_t)Bt * IN), hw((size_t)OUT * IN), hdy((size_t)Bt * OUT);
  std::vector<float> hb(OUT), hxf((size_t)Bt * IN);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xFFFFF) / (float)0xFFFFF * 2.f - 1.f; };
  for (size_t i = 0; i < hx.size(); i++) { const float v = rnd(); hxf[i] = v > 0 ? v : 0.f; hx[i] = f2bf(hxf[i]); }
  for (auto& v : hw) v = f2bf(rnd() / sqrtf((float)IN));
  for (auto& v : hdy) v = f2bf(rnd() / (float)Bt);
  for (auto& v : hb) v = rnd();
  unsigned short *x, *wgt, *dy, *y16, *dx16; float *y, *dx, *bias, *xf;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&wgt, hw.size() * 2)); CK(hipMalloc(&dy, hdy.size() * 2));
  CK(hipMalloc(&y16, (size_t)Bt * OUT * 2)); CK(hipMalloc(&dx16, (size_t)Bt * IN * 2));
  CK(hipMalloc(&y, (size_t)Bt * OUT * 4)); CK(hipMalloc(&dx, (size_t)Bt * IN * 4)); CK(hipMalloc(&bias, OUT * 4)); CK(hipMalloc(&xf, (size_t)Bt * IN * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wgt, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), OUT * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(xf, hxf.data(), hxf.size() * 4, hipMemcpyHostToDevice));
  auto tiles = [](int M, int N) { return (unsigned)(((M + T_BM - 1) / T_BM) * ((N + T_BN - 1) / T_BN)); };
  const double flop = 2.0 * Bt * IN * OUT;
  auto report = [&](const char* nm, float us) { printf("%-44s %8.1f us  %7.1f TFLOP/s (%.3f of 2500)\n", nm, us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0); };

  // forward: y[B][OUT] = relu(x[B][IN] w[OUT][IN]^T + b)
  TArgs a{}; a.A = x; a.B = wgt; a.C = y; a.C16 = y16; a.bias = bias; a.lda = IN; a.ldb = IN; a.ldc = OUT; a.M = Bt; a.N = OUT; a.K = IN; a.act = 1;
  a.a_bytes = (unsigned)((size_t)Bt * IN * 2); a.b_bytes = (unsigned)((size_t)OUT * IN * 2);
  auto k0 = gemm_bf16_2wg_kernel<false, T_EPI_FWD>; set_lds(k0);
  CK(hipMemset(y, 0xFF, (size_t)Bt * OUT * 4));
  hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(256), T_LDS, 0, a);
  CK(hipDeviceSynchronize());
  {
    std::vector<float> hy((size_t)Bt * OUT); std::vector<uint16_t> hy16((size_t)Bt * OUT);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy16.data(), y16, hy16.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; int bad16 = 0;
    for (int smp = 0; smp < 6000; smp++) {
      const int r = (int)((smp * 7919ull + (smp % 3 ? 0 : Bt - 1 - smp % 97)) % Bt), cidx = (int)((smp * 104729ull) % OUT);
      double s = hb[cidx], mass = fabs(hb[cidx]);
      for (int k = 0; k < IN; k++) { const double p = (double)bf2f(hx[(size_t)r * IN + k]) * bf2f(hw[(size_t)cidx * IN + k]); s += p; mass += fabs(p); }
      if (s < 0) s = 0;
      const double e = fabs(hy[(size_t)r * OUT + cidx] - s) / (mass + 1e-30);
      if (e > worst) worst = e;
      if (hy16[(size_t)r * OUT + cidx] != f2bf(hy[(size_t)r * OUT + cidx])) bad16++;
    }
    printf("   fwd check: worst |err| / term mass over 6000 samples = %.3e %s, twin mismatches %d\n", worst, worst < 1e-5 ? "ok" : "WRONG", bad16);
  }
  report("fwd  (kc,kc) fp32 + twin out", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(256), T_LDS, 0, a); }, iters));
  if (diag) {
#define DIAGRUN(D, NAME) { auto kd = gemm_bf16_2wg_kernel<false, T_EPI_FWD, D>; set_lds(kd); \
      report("fwd  " NAME, time_it([&] { hipLaunchKernelGGL(kd, dim3(tiles(a.M, a.N)), dim3(256), T_LDS, 0, a); }, iters)); }
    DIAGRUN(1, "no DMA in the loop (wrong)") DIAGRUN(2, "no fragment reads (wrong)") DIAGRUN(3, "no MFMAs (wrong)") DIAGRUN(4, "no epilogue (wrong)")
    TArgs a2 = a; a2.C16 = nullptr;
    report("fwd  fp32 output only", time_it([&] { hipLaunchKernelGGL(k0, dim3(tiles(a.M, a.N)), dim3(256), T_LDS, 0, a2); }, iters));
  }

  // dX: dx[B][IN] = (dy[B][OUT] w[OUT][IN]) masked by x > 0
  TArgs b{}; b.A = dy; b.B = wgt; b.C = dx; b.C16 = dx16; b.mask = xf; b.mask16 = x; b.ldmask = IN; b.lda = OUT; b.ldb = IN; b.ldc = IN; b.M = Bt; b.N = IN; b.K = OUT;
  b.a_bytes = (unsigned)((size_t)Bt * OUT * 2); b.b_bytes = (unsigned)((size_t)OUT * IN * 2);
  auto k1 = gemm_bf16_2wg_kernel<true, T_EPI_DX>; set_lds(k1);
  CK(hipMemset(dx, 0xFF, (size_t)Bt * IN * 4));
  hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(256), T_LDS, 0, b);
  CK(hipDeviceSynchronize());
  {
    std::vector<float> hd((size_t)Bt * IN);
    CK(hipMemcpy(hd.data(), dx, hd.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int smp = 0; smp < 6000; smp++) {
      const int r = (int)((smp * 7919ull + (smp % 3 ? 0 : Bt - 1 - smp % 97)) % Bt), cidx = (int)((smp * 104729ull) % IN);
      double s = 0, mass = 0;
      for (int k = 0; k < OUT; k++) { const double p = (double)bf2f(hdy[(size_t)r * OUT + k]) * bf2f(hw[(size_t)k * IN + cidx]); s += p; mass += fabs(p); }
      if (!(hxf[(size_t)r * IN + cidx] > 0)) s = 0;
      const double e = fabs(hd[(size_t)r * IN + cidx] - s) / (mass + 1e-30);
      if (e > worst) worst = e;
    }
    printf("   dX  check: worst |err| / term mass over 6000 samples = %.3e %s\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  }
  report("dX   (kc,kr) fp32 + twin out, relu mask", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(256), T_LDS, 0, b); }, iters));
  { TArgs b2 = b; b2.C16 = nullptr; report("dX   fp32 output only", time_it([&] { hipLaunchKernelGGL(k1, dim3(tiles(b.M, b.N)), dim3(256), T_LDS, 0, b2); }, iters)); }
  return 0;
}
