// graph_replay_probe.hip -- minimal reproducer of what the DLRM driver measures on small steps: a step of short kernels on TWO streams (a fork
// and a join through events, as the embedding branch beside the bottom MLP) costs MORE per iteration replayed from a hipGraph than launched
// eagerly (Kaggle shape: 202 vs 169 us per step, profiles/r06_kaggle_{graph,eager}_step_timeline.txt) -- the reason the driver's timed loop
// measures both forms and keeps the faster one (FFConfig::trace_mode).  Stand-alone, no library: to be attached to an
// upstream report.
//
//   step(i):  main stream:  k0 k1 [fork] k2 k3 k4 k5 [join] k6 ... k(N-1)         side stream:  [wait fork] s0 s1 s2 [record join]
//   every kernel spins for `us` microseconds on one workgroup per CU (so that the two branches CAN run side by side).
// Reports, per iteration: eager launches; the same captured once and replayed (hipGraphLaunch); the same with the side branch on the main
// stream (one-stream graph).  If replay linearises the branches its time is ~ (N + 3) * us + floor; eager is ~ N * us.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/graph_replay_probe.hip -o tools/lab/graph_replay_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (ticks < 0) *sink = 1;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 12;          // kernels on the main stream
  const double us = argc > 2 ? atof(argv[2]) : 10.0;    // duration of each
  const int grid = argc > 3 ? atoi(argv[3]) : 128;      // workgroups per kernel (half the CUs: both branches fit)
  const int iters = 300;
  int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
  const long long ticks = (long long)(us * 1e-3 * wc_khz);
  int* sink; CK(hipMalloc(&sink, 4));
  hipStream_t sm, ss; CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
  hipEvent_t fork, join, t0, t1; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto k = [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(64), 0, s, ticks, sink); };
  auto step = [&](bool two_streams) {
    hipStream_t side = two_streams ? ss : sm;
    k(sm); k(sm);
    if (two_streams) { CK(hipEventRecord(fork, sm)); CK(hipStreamWaitEvent(ss, fork, 0)); }
    k(side); k(side); k(side);
    if (two_streams) CK(hipEventRecord(join, ss));
    for (int i = 2; i < 6 && i < N; i++) k(sm);
    if (two_streams) CK(hipStreamWaitEvent(sm, join, 0));
    for (int i = 6; i < N; i++) k(sm);
  };
  auto time_loop = [&](auto&& body) {
    for (int i = 0; i < 50; i++) body();
    CK(hipStreamSynchronize(sm));
    CK(hipEventRecord(t0, sm));
    for (int i = 0; i < iters; i++) body();
    CK(hipEventRecord(t1, sm)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    return ms * 1000.f / iters;
  };
  printf("%d kernels of %.1f us on the main stream, 3 on a forked side stream beside kernels 2..5, %d workgroups each\n", N, us, grid);
  printf("  ideal, branches side by side: %.1f us per step; serialised: %.1f us\n", N * us, (N + 3) * us);
  for (int two = 1; two >= 0; two--) {
    const float eager = time_loop([&] { step(two); });
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(sm, hipStreamCaptureModeThreadLocal));
    step(two);
    CK(hipStreamEndCapture(sm, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float replay = time_loop([&] { CK(hipGraphLaunch(ge, sm)); });
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    printf("  %s: eager %.1f us per step, hipGraph replay %.1f us per step (%zu nodes)  -> replay %+.1f us\n", two ? "two streams" : "one stream ", eager, replay, nn, replay - eager);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
