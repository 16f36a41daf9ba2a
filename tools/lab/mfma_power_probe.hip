// mfma_power_probe.hip -- what the matrix pipe sustains on bare MFMAs, by instruction shape, with every SIMD of the chip busy (two waves per SIMD):
// v_mfma_f32_16x16x32_bf16 (what the bf16-pipe GEMMs of this repo issue) against v_mfma_f32_32x32x16_bf16 (same flops per cycle on paper, half the
// operand-register reads per flop).  A GEMM at ~86 % pipe-busy runs power-limited: if one shape drew less power per flop, it
// would sustain a higher clock.  Operands: random bf16 values (the toggle rate matters for power), accumulators kept, no memory in the loop.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/mfma_power_probe.hip -o tools/lab/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_kernel(int iters, const s16x8* __restrict__ src, float* sink) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + i) & 4095]); b[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + 4 + i) & 4095]); }
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);      // 16 x 16 KFLOP... 16 MFMAs of 16384 flop
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 2; r++)                  // 8 MFMAs of 32768 flop: the same flops per iteration as the 16 above
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * r + i], b[2 * r + j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) s += acc[i][j][0] + acc[i][j][15];
  }
  if (s == 123.456f) *sink = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  s16x8* src; float* sink; CK(hipMalloc(&src, 4096 * 16)); CK(hipMalloc(&sink, 4));
  { short h[4096 * 8]; unsigned st = 12345; for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (short)(0x3C00 + ((st >> 9) & 0x7FF) * ((st >> 31) ? 1 : -1)); } CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice)); }
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int round = 0; round < 2; round++)
    for (int shape : {16, 32}) {
      auto launch = [&] { if (shape == 16) hipLaunchKernelGGL(mfma_kernel<16>, dim3(256), dim3(512), 0, 0, iters, src, sink); else hipLaunchKernelGGL(mfma_kernel<32>, dim3(256), dim3(512), 0, 0, iters, src, sink); };
      for (int i = 0; i < 6; i++) launch();          // ~150 ms: clocks settled
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0)); for (int i = 0; i < 5; i++) launch(); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      const double flop = 5.0 * 256 * 8 * (double)iters * 16 * 16384;
      printf("%s: %.1f TFLOP/s bf16 (%.3f of 2500) = %.1f fp32-equivalent at six products\n", shape == 16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", flop / ms * 1e-9, flop / ms * 1e-9 / 2500, flop / ms * 1e-9 / 6);
    }
  return 0;
}
