// corun_probe.hip -- what slows a small kernel down beside a persistent GEMM?  (lab tool)
// Victims: (a) an index-heavy loop (ballots, LDS atomics, barriers -- the window phase of the table update), (b) a chain of dependent
// random 512-byte row reads (its reduce phase).  Neighbours, one 256-thread workgroup per CU each: (1) a wave per SIMD issuing
// v_mfma_f32_16x16x4_f32 back to back, no memory at all; (2) a streamer reading a 1 GB buffer with 16-byte loads, no MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/corun_probe.hip -o tools/lab/corun_probe && tools/lab/corun_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 1) void mfma_spin(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  f32x4 s = acc[0];
  for (int i = 1; i < 8; i++) s += acc[i];
  if (s.x == 123.456f) out[0] = s.y;
}

// the same with the wave stepping aside between MFMAs: s_nop (the wave's own idle cycles, not the VALU port) for most of the 32 cycles the
// matrix pipe needs per 16x16x4 fp32 MFMA
template <int PAD>
__global__ __launch_bounds__(256, 1) void mfma_spin_pad(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        if (PAD >= 16) asm volatile("s_nop 15" ::: "memory");
        if (PAD >= 32) asm volatile("s_nop 15" ::: "memory");
        if (PAD % 16) asm volatile("s_nop %0" :: "n"((PAD % 16) - 1) : "memory");
      }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  f32x4 s = acc[0];
  for (int i = 1; i < 8; i++) s += acc[i];
  if (s.x == 123.456f) out[0] = s.y;
}

__global__ __launch_bounds__(256, 1) void streamer(const float4* buf, size_t n4, int rounds, float* out) {
  float4 s = make_float4(0, 0, 0, 0);
  for (int r = 0; r < rounds; r++)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256 * 4) {
      float4 v0 = buf[i], v1 = buf[(i + (size_t)gridDim.x * 256) % n4], v2 = buf[(i + 2 * (size_t)gridDim.x * 256) % n4], v3 = buf[(i + 3 * (size_t)gridDim.x * 256) % n4];
      s.x += v0.x + v1.x + v2.x + v3.x;
    }
  if (s.x == 123.456f) out[0] = s.x;
}

// (a) index-heavy: per iteration 8 ballots, an LDS atomic, a barrier
__global__ __launch_bounds__(256, 4) void victim_index(unsigned* out, int iters, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  __shared__ unsigned h[512];
  for (int i = threadIdx.x; i < 512; i += 256) h[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const bool one = (x >> bit) & 1u;
      const unsigned long long bal = __ballot(one);
      peers &= one ? bal : ~bal;
    }
    acc += __popcll(peers);
    atomicAdd(&h[x & 511u], 1u);
    x = x * 1664525u + 1013904223u;
    __syncthreads();
  }
  if (acc == 0xdeadbeefu) out[0] = h[0];
}

// (b) latency chain: each lane group of 32 reads a random 512-byte row, the next row's index depends on it
__global__ __launch_bounds__(256, 4) void victim_rows(const float4* table, unsigned rows, unsigned* out, int iters, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  unsigned r = (blockIdx.x * 8 + (threadIdx.x >> 5)) * 2654435761u % rows;
  float s = 0.f;
  for (int it = 0; it < iters; it++) {
    const float4 v = table[(size_t)r * 32 + (threadIdx.x & 31)];
    s += v.x;
    r = (r * 1664525u + 1013904223u + (unsigned)(v.y != 0.12345f)) % rows;
  }
  if (s == 123.456f) out[0] = (unsigned)s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* fo; unsigned* uo; CK(hipMalloc(&fo, 64)); CK(hipMalloc(&uo, 64));
  const size_t n4 = (size_t)1 << 26;        // 1 GB
  float4* big; CK(hipMalloc(&big, n4 * 16)); CK(hipMemset(big, 0, n4 * 16));
  const unsigned rows = 4000000;            // 2 GB of 512-byte rows
  float4* table; CK(hipMalloc(&table, (size_t)rows * 512)); CK(hipMemset(table, 0, (size_t)rows * 512));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto victim = [&](int which, int prio) {
    if (which == 0) hipLaunchKernelGGL(victim_index, dim3(cus * 4), dim3(256), 0, s2, uo, 400, prio);
    else hipLaunchKernelGGL(victim_rows, dim3(cus * 4), dim3(256), 0, s2, table, rows, uo, 40, prio);
  };
  auto neighbour = [&](int which) {
    if (which == 1) hipLaunchKernelGGL(mfma_spin, dim3(cus), dim3(256), 0, s1, fo, 3000);          // ~ 3000 * 32 MFMAs * 32 cycles = 1.3 ms
    if (which == 2) hipLaunchKernelGGL(streamer, dim3(cus), dim3(256), 0, s1, big, n4, 6, fo);
    if (which == 3) hipLaunchKernelGGL(mfma_spin_pad<16>, dim3(cus), dim3(256), 0, s1, fo, 3000);
    if (which == 4) hipLaunchKernelGGL(mfma_spin_pad<24>, dim3(cus), dim3(256), 0, s1, fo, 3000);
    if (which == 5) hipLaunchKernelGGL(mfma_spin_pad<28>, dim3(cus), dim3(256), 0, s1, fo, 3000);
    if (which == 6) hipLaunchKernelGGL(mfma_spin_pad<32>, dim3(cus), dim3(256), 0, s1, fo, 3000);
  };
  const char* vn[2] = {"index-heavy (ballots, LDS atomics, barriers)", "dependent random 512-byte row reads"};
  const char* nn[7] = {"alone", "beside MFMA-only waves (one per SIMD, no memory)", "beside a streaming reader (no MFMA)", "beside MFMA waves + s_nop 16 after each",
                       "beside MFMA waves + s_nop 24", "beside MFMA waves + s_nop 28", "beside MFMA waves + s_nop 32"};
  for (int v = 0; v < 2; v++)
    for (int prio = 0; prio < 2; prio++)
    for (int n = 0; n < 7; n++) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        neighbour(n);
        if (n) { for (volatile int spin = 0; spin < 200000; spin++) {} }      /* the victim starts once the neighbour is on the chip */
        CK(hipEventRecord(e0, s2));
        victim(v, prio);
        CK(hipEventRecord(e1, s2));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        CK(hipDeviceSynchronize());
      }
      printf("%-46s prio %d  %-52s %8.1f us\n", vn[v], prio ? 3 : 0, nn[n], best * 1e3f);
    }
  // the neighbours' own durations, for scale
  for (int n = 1; n < 7; n++) {
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, s1)); neighbour(n); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("neighbour %d alone: %.1f us\n", n, ms * 1e3f);
  }
  return 0;
}
