// store_probe.hip -- lab: what a GEMM epilogue's store pattern costs by itself (no arithmetic).  One workgroup of 8 waves per 256 x 256 tile of a
// 32768 x N matrix, each wave "stores" its 2 x (64 rows x 64 columns) blocks in the chosen pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/store_probe.hip -o tools/lab/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: plane image, whole lines: a row's 64 columns = 384 contiguous bytes, rows `pitch` apart; 24 instructions per block
// mode 1: the same bytes as 1 KB contiguous per instruction, a wave's blocks back to back
// mode 2: fp32 image: 256 contiguous bytes per row, 4 rows per instruction, 16 instructions per block
// mode 3: plane image as 64-byte pieces (16 rows x 64 B per instruction: 3 instructions per (16 rows, 32 columns))
// mode 4: mode 0 with nontemporal stores
template <int MODE>
__global__ __launch_bounds__(512) void probe(char* out, int64_t pitch, int nbx, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2, wc = wave & 3;
  const unsigned total = gridDim.x, w = blockIdx.x;
  const unsigned xcd = w & 7u, loc = w >> 3, qq = total >> 3;
  const unsigned tile = xcd * qq + loc;
  const int by = tile / nbx, bx = tile % nbx;
  const uint4 val = make_uint4(tile, wave, lane, 7);
  for (int rep = 0; rep < reps; rep++)
  for (int half = 0; half < 2; half++) {
    const int64_t rowb = (int64_t)by * 256 + half * 128 + grp * 64;
    if (MODE == 0 || MODE == 4) {
      char* base = out + rowb * pitch + (int64_t)(bx * 4 + wc) * 384;
#pragma unroll 4
      for (int it = 0; it < 24; it++) {
        const int L = it * 64 + lane, r = L / 24, ch = L - r * 24;
        uint4* p = reinterpret_cast<uint4*>(base + r * pitch + ch * 16);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if (MODE == 4) __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(p)); else *p = val;
      }
    } else if (MODE == 1) {
      char* base = out + ((int64_t)(tile * 8 + wave) * 2 + half) * 24576;
#pragma unroll 4
      for (int it = 0; it < 24; it++) *reinterpret_cast<uint4*>(base + it * 1024 + lane * 16) = val;
    } else if (MODE == 2) {
      char* base = out + rowb * pitch + (int64_t)(bx * 4 + wc) * 256;
#pragma unroll 4
      for (int it = 0; it < 16; it++) *reinterpret_cast<uint4*>(base + (it * 4 + (lane >> 4)) * pitch + (lane & 15) * 16) = val;
    } else if (MODE == 3) {
      char* base = out + rowb * pitch + (int64_t)(bx * 4 + wc) * 384;
#pragma unroll 2
      for (int p = 0; p < 4; p++)
        for (int h = 0; h < 2; h++)
          for (int pl = 0; pl < 3; pl++)
            *reinterpret_cast<uint4*>(base + (p * 16 + (lane >> 2)) * pitch + h * 192 + pl * 64 + (lane & 3) * 16) = val;
    }
  }
}

static float time_it(const std::function<void()>& f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
  return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
  const int Bt = 32768, N = argc > 1 ? atoi(argv[1]) : 1024;
  const int nbx = N / 256, tiles = (Bt / 256) * nbx;
  char* buf; const size_t bytes = (size_t)Bt * (N * 6 + 1024) + (1 << 20);
  CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  const double pl_bytes = (double)Bt * N * 6, f_bytes = (double)Bt * N * 4;
  auto rep = [&](const char* what, float us, double by) { printf("%-64s %8.1f us  %6.2f TB/s\n", what, us, by / us / 1e6); fflush(stdout); };
  for (int64_t pitch : {(int64_t)N * 6, (int64_t)N * 6 + 128, (int64_t)N * 6 + 256, (int64_t)N * 6 + 512}) {
    char nm[96];
    snprintf(nm, sizeof nm, "planes, whole lines (384 B / row), pitch %ld", (long)pitch);
    rep(nm, time_it([&] { hipLaunchKernelGGL(probe<0>, dim3(tiles), dim3(512), 0, 0, buf, pitch, nbx, 1); }, 20), pl_bytes);
  }
  rep("planes, nontemporal, pitch 6 N", time_it([&] { hipLaunchKernelGGL(probe<4>, dim3(tiles), dim3(512), 0, 0, buf, (int64_t)N * 6, nbx, 1); }, 20), pl_bytes);
  rep("planes as 1 KB contiguous per instruction", time_it([&] { hipLaunchKernelGGL(probe<1>, dim3(tiles), dim3(512), 0, 0, buf, (int64_t)N * 6, nbx, 1); }, 20), pl_bytes);
  rep("planes as 64 B pieces (16 rows per instruction)", time_it([&] { hipLaunchKernelGGL(probe<3>, dim3(tiles), dim3(512), 0, 0, buf, (int64_t)N * 6, nbx, 1); }, 20), pl_bytes);
  rep("fp32 image (256 B / row, 4 rows per instruction), pitch 4 N", time_it([&] { hipLaunchKernelGGL(probe<2>, dim3(tiles), dim3(512), 0, 0, buf, (int64_t)N * 4, nbx, 1); }, 20), f_bytes);
  rep("fp32 image, pitch 4 N + 128", time_it([&] { hipLaunchKernelGGL(probe<2>, dim3(tiles), dim3(512), 0, 0, buf, (int64_t)N * 4 + 128, nbx, 1); }, 20), f_bytes);
  return 0;
}
