// runtime.hip -- context, memory, streams/events/graphs and the fill / RNG kernels of
// libffhip.so.  What Legion/Realm give the reference's operator tasks (device memory
// in regions, a per-task stream [ref: src/runtime/cuda_helper.cu:5-31], trace replay
// [ref: examples/cpp/DLRM/dlrm.cc:174-181]) is exposed here as plain HIP objects.
#include "ffh_common.h"

#include <mutex>
#include <stdlib.h>
#include <new>

extern "C" {

int         ffh_abi_version(void) { return FFH_ABI_VERSION; }
const char* ffh_backend_name(void) { return "hip-gfx950"; }

int ffh_ctx_create(ffh_ctx** out, int device) {
  if (!out) return FFH_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return FFH_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return FFH_ERR_HIP;
  ffh_ctx* c = new (std::nothrow) ffh_ctx();
  if (!c) return FFH_ERR_NOMEM;
  memset(c, 0, sizeof *c);
  c->device = device;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) == hipSuccess) c->num_cus = p.multiProcessorCount;
  if (c->num_cus <= 0) c->num_cus = 256;
  if (hipMalloc((void**)&c->zeros, 256) != hipSuccess || hipMemset(c->zeros, 0, 256) != hipSuccess) { (void)hipGetLastError(); c->zeros = nullptr; }
  else (void)hipStreamSynchronize(nullptr);      // the null-stream memset is not ordered against the non-blocking streams the callers create: wait for it once, here
  *out = c;
  return FFH_OK;
}

// ffh_ctx_default: one library-owned ctx per device, for static call sites that get no handle
int ffh_ctx_default(ffh_ctx** out) {
  static std::mutex mu;
  static ffh_ctx* per_device[64];
  if (!out) return FFH_ERR_BAD_ARG;
  *out = nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return FFH_ERR_HIP; }
  std::lock_guard<std::mutex> lock(mu);
  if (!per_device[dev]) {
    const int rc = ffh_ctx_create(&per_device[dev], dev);
    if (rc != FFH_OK) return rc;
  }
  *out = per_device[dev];
  return FFH_OK;
}

static void ffh_scratch_free(ffh_ctx* c, int i) {
  auto& t = c->scratch[i];
  if (t.sk_slots) (void)hipFree(t.sk_slots);
  if (t.sk_cnt) (void)hipFree(t.sk_cnt);
  if (t.skinny_ws) (void)hipFree(t.skinny_ws);
  if (t.skinny_cnt) (void)hipFree(t.skinny_cnt);
  if (t.x3_slots) (void)hipFree(t.x3_slots);
  t = {};
}

// ffh_ctx_reserve_scratch (ABI 12): the per-stream scratch of the forms that meet through memory -- stream-K with fix-up (linear_sk.hip),
// the narrow-layer backward's last-arriver sums (linear.hip).  The one place they are allocated: the compute entry points look a
// stream's set up and run the other forms when there is none.  Like LinearMeta's ones vector [ref: src/ops/linear.cu:986-994], but
// released (ffh_stream_destroy, ffh_ctx_destroy).  Idempotent; not during a stream capture (hipMalloc may synchronise).
int ffh_ctx_reserve_scratch(ffh_ctx* c, ffh_stream st) {
  if (!c) return FFH_ERR_BAD_ARG;
  hipStream_t s = as_stream(st);
  for (int i = 0; i < c->nscratch; i++)
    if (c->scratch[i].stream == (void*)s) return FFH_OK;
  if (c->nscratch >= FFH_MAX_SCRATCH_STREAMS) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "ctx_reserve_scratch: FFH_MAX_SCRATCH_STREAMS streams hold scratch already (destroy one)");
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if (cap != hipStreamCaptureStatusNone) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "ctx_reserve_scratch: the stream is being captured");
  const size_t G = (size_t)(c->num_cus & ~7);
  auto& t = c->scratch[c->nscratch];
  t = {};
  t.stream = (void*)s;
  bool ok = hipMalloc((void**)&t.sk_slots, 2 * G * kSkTileFloats * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&t.sk_cnt, (G + 16) * sizeof(unsigned)) == hipSuccess &&
            hipMalloc((void**)&t.skinny_ws, (size_t)kSkinnyWsBlocks * kSkinnyWsRow * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&t.skinny_cnt, 64) == hipSuccess &&
            hipMalloc((void**)&t.x3_slots, (size_t)kX3DwSlots * kX3TileFloats * sizeof(float)) == hipSuccess;
  // (counters cleared ON the stream the launches go to: a null-stream hipMemset is not ordered against a non-blocking stream, and a
  //  launch that finds a counter mid-way never sees its last arriver)
  ok = ok && hipMemsetAsync(t.sk_cnt, 0, (G + 16) * sizeof(unsigned), s) == hipSuccess && hipMemsetAsync(t.skinny_cnt, 0, 64, s) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    ffh_scratch_free(c, c->nscratch);
    return ffh_fail(c, FFH_ERR_NOMEM, "ctx_reserve_scratch: out of device memory");
  }
  c->nscratch++;
  return FFH_OK;
}

int ffh_ctx_destroy(ffh_ctx* c) {
  if (c && c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c && c->zeros) (void)hipFree(c->zeros);
  for (int i = 0; c && i < c->nscratch; i++) ffh_scratch_free(c, i);
  delete c;
  return FFH_OK;
}

const char* ffh_last_error_string(const ffh_ctx* c) { return c ? c->err : "null ctx"; }
const char* ffh_linear_last_route(const ffh_ctx* c) { return c ? c->route : ""; }
const char* ffh_embedding_last_route(const ffh_ctx* c) { return c ? c->emb_route : ""; }

int ffh_device_query(ffh_ctx* c, ffh_device_info* info) {
  if (!c || !info) return FFH_ERR_BAD_ARG;
  memset(info, 0, sizeof *info);
  hipDeviceProp_t p;
  FFH_HIP_TRY(c, hipGetDeviceProperties(&p, c->device));
  snprintf(info->name, sizeof info->name, "%s", p.name);
  snprintf(info->arch, sizeof info->arch, "%s", p.gcnArchName);
  info->compute_units = p.multiProcessorCount;
  info->wavefront_size = p.warpSize;
  info->total_mem_bytes = (int64_t)p.totalGlobalMem;
  info->lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
  info->clock_khz = p.clockRate;
  return FFH_OK;
}

int ffh_ctx_set_workspace(ffh_ctx* c, void* ws, size_t bytes) {
  if (!c) return FFH_ERR_BAD_ARG;
  c->ws = ws;
  c->ws_bytes = bytes;
  return FFH_OK;
}

int ffh_ctx_set_math_mode(ffh_ctx* c, int mode) {
  if (!c || (mode != FFH_MATH_DEFAULT && mode != FFH_MATH_TENSOR_OP_BF16 && mode != FFH_MATH_FP32_SPLIT_BF16X3 && mode != FFH_MATH_FP32_SPLIT_BF16X3_ALL)) return FFH_ERR_BAD_ARG;
  c->math_mode = mode;
  return FFH_OK;
}

int ffh_ctx_set_dw_cu_reserve(ffh_ctx* c, int ncus) {
  if (!c || ncus < 0 || ncus >= c->num_cus) return FFH_ERR_BAD_ARG;
  c->dw_cu_reserve = ncus / 8 * 8;
  return FFH_OK;
}

static int mirror_set(ffh_ctx* c, const void* base, size_t bytes, void* twin, int planes, unsigned align) {
  if (!c || !base || bytes == 0 || ((uintptr_t)base & (align - 1)) || ((uintptr_t)twin & (align - 1))) return FFH_ERR_BAD_ARG;
  int at = -1;
  for (int i = 0; i < c->nmirrors; i++) if (c->mirrors[i].base == (const char*)base && c->mirrors[i].planes == planes) at = i;
  if (!twin) {
    if (at >= 0) { c->mirrors[at] = c->mirrors[c->nmirrors - 1]; c->nmirrors--; }
    return FFH_OK;
  }
  if (at < 0) {
    if (c->nmirrors >= kMirrorRegions) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "mirror_set: more than 64 regions");
    at = c->nmirrors++;
  }
  c->mirrors[at].base = (const char*)base; c->mirrors[at].bytes = bytes; c->mirrors[at].twin = (char*)twin; c->mirrors[at].planes = planes;
  return FFH_OK;
}
int ffh_ctx_bf16_mirror_set(ffh_ctx* c, const void* base, size_t bytes, void* twin) { return mirror_set(c, base, bytes, twin, 1, 16); }
int ffh_ctx_bf16x3_mirror_set(ffh_ctx* c, const void* base, size_t bytes, void* planes) { return mirror_set(c, base, bytes, planes, 3, 128); }

__global__ __launch_bounds__(256) void convert_bf16_kernel(unsigned short* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && (((uintptr_t)(src + i)) & 15) == 0 && (((uintptr_t)(dst + i)) & 7) == 0) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      const bf2 lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<uint2*>(dst + i) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    } else {
      for (int k = 0; k < 4 && i + k < n; k++) { const __bf16 b = (__bf16)src[i + k]; dst[i + k] = __builtin_bit_cast(unsigned short, b); }
    }
  }
}

int ffh_convert_f32_to_bf16(ffh_ctx* c, void* dst, const float* src, int64_t n, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && ((dst && src) || n == 0), "convert_f32_to_bf16: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL(convert_bf16_kernel, dim3(ffh_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(s), (unsigned short*)dst, src, n);
  FFH_LAUNCH_CHECK(c, "convert_bf16_kernel");
  return FFH_OK;
}

// fp32 -> three bf16 terms, into the I32 image (ff_hip.h).  `img` is the image of the region, e0 the index of src[0] in it; element (r, c) of
// the sub-matrix is region element e0 + r * ld + c.  Four elements per lane where the pieces are 16-byte aligned on both sides.
__global__ __launch_bounds__(256) void convert_bf16x3_kernel(char* __restrict__ img, const float* __restrict__ src, int64_t e0, int64_t rows, int64_t cols, int64_t ld, int vec) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t per_row = cols / 4, total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
      const int64_t r = i / per_row, cc = (i - r * per_row) * 4;
      const float4 v = *reinterpret_cast<const float4*>(src + r * ld + cc);
      uint2 p1, p2, p3;
      ffh_split_bf16x3(v, p1, p2, p3);
      const int64_t e = e0 + r * ld + cc;
      char* d = img + (e >> 5) * 192 + (e & 31) * 2;
      *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + 64) = p2; *reinterpret_cast<uint2*>(d + 128) = p3;
    }
  } else {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
      const int64_t r = i / cols, cc = i - r * cols;
      const float x = src[r * ld + cc];
      const __bf16 b1 = (__bf16)x; const float r1 = x - (float)b1;
      const __bf16 b2 = (__bf16)r1; const float r2 = r1 - (float)b2;
      const __bf16 b3 = (__bf16)r2;
      const int64_t e = e0 + r * ld + cc;
      unsigned short* d = reinterpret_cast<unsigned short*>(img + (e >> 5) * 192 + (e & 31) * 2);
      d[0] = __builtin_bit_cast(unsigned short, b1); d[32] = __builtin_bit_cast(unsigned short, b2); d[64] = __builtin_bit_cast(unsigned short, b3);
    }
  }
}

int ffh_convert_f32_to_bf16x3(ffh_ctx* c, const float* src, int64_t rows, int64_t cols, int64_t ld, ffh_stream s) {
  FFH_REQUIRE(c, rows >= 0 && cols >= 0 && (rows <= 1 || ld >= cols), "convert_f32_to_bf16x3: bad dims");
  if (rows == 0 || cols == 0) return FFH_OK;
  FFH_REQUIRE(c, src != nullptr, "convert_f32_to_bf16x3: null pointer");
  int col0 = 0;
  char* grp = ffh_planes_of(c, src, (size_t)((rows - 1) * ld + cols) * 4, &col0, true);
  if (!grp) return ffh_fail(c, FFH_ERR_BAD_ARG, "convert_f32_to_bf16x3: not inside a region registered with ffh_ctx_bf16x3_mirror_set");
  // (the kernel takes the group's address as the image origin and col0 as e0: the same addresses as from the region's base)
  const int vec = (col0 % 4 == 0) && (cols % 4 == 0) && (rows == 1 || ld % 4 == 0) && (((uintptr_t)src & 15) == 0);
  const int64_t work = vec ? rows * (cols / 4) : rows * cols;
  hipLaunchKernelGGL(convert_bf16x3_kernel, dim3(ffh_grid(work, 256)), dim3(256), 0, as_stream(s), grp, src, (int64_t)col0, rows, cols, ld, vec);
  FFH_LAUNCH_CHECK(c, "convert_bf16x3_kernel");
  return FFH_OK;
}

int ffh_ctx_set_deterministic(ffh_ctx* c, int on) {
  if (!c) return FFH_ERR_BAD_ARG;
  c->deterministic = on ? 1 : 0;
  return FFH_OK;
}

int ffh_malloc(ffh_ctx* c, void** p, size_t bytes) {
  if (!c || !p) return FFH_ERR_BAD_ARG;
  *p = nullptr;
  hipError_t e = hipMalloc(p, bytes ? bytes : 256);
  if (e != hipSuccess) { ffh_fail_hip(c, e, "hipMalloc"); return FFH_ERR_NOMEM; }
  return FFH_OK;
}
int ffh_free(ffh_ctx* c, void* p) { if (p) FFH_HIP_TRY(c, hipFree(p)); return FFH_OK; }
int ffh_memcpy_h2d(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) {
  FFH_HIP_TRY(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, as_stream(st))); return FFH_OK; }
int ffh_memcpy_d2h(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) {
  FFH_HIP_TRY(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, as_stream(st))); return FFH_OK; }
int ffh_memcpy_d2d(ffh_ctx* c, void* d, const void* s, size_t n, ffh_stream st) {
  FFH_HIP_TRY(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, as_stream(st))); return FFH_OK; }

int ffh_stream_create(ffh_ctx* c, ffh_stream* s) {
  if (!s) return FFH_ERR_BAD_ARG;
  hipStream_t st;
  // A/B switch (tools/ab.sh): FFH_STREAM_PRIOS="p0,p1,p2" gives the n-th stream this process creates HIP priority pn
#ifdef FFH_LAB
  static int created = 0;
  const char* pr = getenv("FFH_STREAM_PRIOS");
  if (pr) {
    int idx = created++, prio = 0;
    const char* p = pr;
    for (int i = 0; i <= idx && p && *p; i++) { prio = atoi(p); p = strchr(p, ','); if (p) p++; else if (i < idx) { prio = 0; break; } }
    FFH_HIP_TRY(c, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
    *s = (ffh_stream)st;
    return FFH_OK;
  }
#endif
  FFH_HIP_TRY(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = (ffh_stream)st;
  return FFH_OK;
}
int ffh_stream_create_with_priority(ffh_ctx* c, ffh_stream* s, int priority) {
  if (!s) return FFH_ERR_BAD_ARG;
  int lo = 0, hi = 0;                                     // numerically: hi <= lo, hi is the greatest priority
  FFH_HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
  if (priority < hi) priority = hi;
  if (priority > lo) priority = lo;
  hipStream_t st;
  FFH_HIP_TRY(c, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
  *s = (ffh_stream)st;
  return FFH_OK;
}
int ffh_stream_destroy(ffh_ctx* c, ffh_stream s) {
  if (!s) return FFH_OK;
  // the stream's scratch goes with it (its kernels must have finished before the buffers are freed: hipFree waits for the device)
  for (int i = 0; c && i < c->nscratch; i++)
    if (c->scratch[i].stream == (void*)s) {
      (void)hipStreamSynchronize(as_stream(s));
      ffh_scratch_free(c, i);
      c->scratch[i] = c->scratch[c->nscratch - 1];
      c->scratch[c->nscratch - 1] = {};
      c->nscratch--;
      break;
    }
  FFH_HIP_TRY(c, hipStreamDestroy(as_stream(s)));
  return FFH_OK;
}
int ffh_stream_sync(ffh_ctx* c, ffh_stream s) { FFH_HIP_TRY(c, hipStreamSynchronize(as_stream(s))); return FFH_OK; }
int ffh_device_sync(ffh_ctx* c) { FFH_HIP_TRY(c, hipDeviceSynchronize()); return FFH_OK; }

int ffh_event_create(ffh_ctx* c, ffh_event* e) {
  if (!e) return FFH_ERR_BAD_ARG;
  hipEvent_t ev;
  FFH_HIP_TRY(c, hipEventCreate(&ev));
  *e = (ffh_event)ev;
  return FFH_OK;
}
int ffh_event_create_sync(ffh_ctx* c, ffh_event* e) {
  if (!e) return FFH_ERR_BAD_ARG;
  hipEvent_t ev;
  FFH_HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  *e = (ffh_event)ev;
  return FFH_OK;
}
int ffh_event_destroy(ffh_ctx* c, ffh_event e) { if (e) FFH_HIP_TRY(c, hipEventDestroy((hipEvent_t)e)); return FFH_OK; }
int ffh_event_record(ffh_ctx* c, ffh_event e, ffh_stream s) { FFH_HIP_TRY(c, hipEventRecord((hipEvent_t)e, as_stream(s))); return FFH_OK; }
int ffh_event_sync(ffh_ctx* c, ffh_event e) { FFH_HIP_TRY(c, hipEventSynchronize((hipEvent_t)e)); return FFH_OK; }
int ffh_stream_wait_event(ffh_ctx* c, ffh_stream s, ffh_event e) { FFH_HIP_TRY(c, hipStreamWaitEvent(as_stream(s), (hipEvent_t)e, 0)); return FFH_OK; }
int ffh_event_elapsed_ms(ffh_ctx* c, ffh_event a, ffh_event b, float* ms) {
  if (!ms) return FFH_ERR_BAD_ARG;
  FFH_HIP_TRY(c, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return FFH_OK;
}

// The reference memoises its task graph with Legion traces (begin_trace/end_trace(111),
// [ref: examples/cpp/DLRM/dlrm.cc:174-181]); the MI355X counterpart is a captured hipGraph.
int ffh_graph_begin_capture(ffh_ctx* c, ffh_stream s) {
  FFH_HIP_TRY(c, hipStreamBeginCapture(as_stream(s), hipStreamCaptureModeThreadLocal));
  return FFH_OK;
}
int ffh_graph_end_capture(ffh_ctx* c, ffh_stream s, ffh_graph* g) {
  if (!g) return FFH_ERR_BAD_ARG;
  hipGraph_t graph = nullptr;
  FFH_HIP_TRY(c, hipStreamEndCapture(as_stream(s), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return ffh_fail_hip(c, e, "hipGraphInstantiate");
  *g = (ffh_graph)exec;
  return FFH_OK;
}
int ffh_graph_launch(ffh_ctx* c, ffh_graph g, ffh_stream s) {
  FFH_HIP_TRY(c, hipGraphLaunch((hipGraphExec_t)g, as_stream(s)));
  return FFH_OK;
}
int ffh_graph_destroy(ffh_ctx* c, ffh_graph g) { if (g) FFH_HIP_TRY(c, hipGraphExecDestroy((hipGraphExec_t)g)); return FFH_OK; }

}  // extern "C"

// ---------------------------------------------------------------------------
// fill / RNG kernels: pure streaming stores, 16 B per lane
// ---------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, int64_t n, float v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4 v4 = make_float4(v, v, v, v);
  for (int64_t k = i; k < n4; k += stride) p4[k] = v4;
  for (int64_t k = (n4 << 2) + i; k < n; k += stride) p[k] = v;
}

template <int MODE>  // 0 uniform(lo,hi), 1 u24 in [0,1), 2 bernoulli
__global__ __launch_bounds__(256) void gen_f32_kernel(float* __restrict__ p, int64_t n, uint64_t seed, int64_t first, float lo, float hi) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t h = ffh_hash(seed, (uint64_t)(first + i));
    float v;
    if (MODE == 0) v = ffh_uniform(h, lo, hi);
    else if (MODE == 1) v = ffh_u24(h);
    else v = ffh_bernoulli(h);
    p[i] = v;
  }
}

__global__ __launch_bounds__(256) void gen_idx_kernel(int64_t* __restrict__ p, int64_t n, uint64_t seed, int64_t first, int64_t R) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = ffh_index(ffh_hash(seed, (uint64_t)(first + i)), R);
}

}  // namespace

extern "C" {

// assign_kernel [ref: src/runtime/cuda_helper.cu:52-60]
int ffh_fill_f32(ffh_ctx* c, float* p, int64_t n, float v, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && (p || n == 0), "fill_f32: bad args");
  if (n == 0) return FFH_OK;
  if (((uintptr_t)p & 15) != 0) {  // unaligned base: plain memset pattern through the scalar tail path
    hipLaunchKernelGGL((gen_f32_kernel<0>), dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), p, n, 0, 0, v, v);
  } else {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(ffh_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(s), p, n, v);
  }
  FFH_LAUNCH_CHECK(c, "fill_f32");
  return FFH_OK;
}

// ZeroInitializer::init_task / Op::zero_grad [ref: src/runtime/initializer_kernel.cu:209-241, src/runtime/model.cc:466-490]
int ffh_zero(ffh_ctx* c, void* p, size_t bytes, ffh_stream s) {
  if (bytes == 0) return FFH_OK;
  FFH_HIP_TRY(c, hipMemsetAsync(p, 0, bytes, as_stream(s)));
  return FFH_OK;
}

int ffh_init_uniform(ffh_ctx* c, float* p, int64_t n, uint64_t seed, float lo, float hi, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && (p || n == 0), "init_uniform: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL((gen_f32_kernel<0>), dim3(ffh_grid(n, 256, 8192)), dim3(256), 0, as_stream(s), p, n, seed, (int64_t)0, lo, hi);
  FFH_LAUNCH_CHECK(c, "init_uniform");
  return FFH_OK;
}
int ffh_gen_indices(ffh_ctx* c, int64_t* p, int64_t n, uint64_t seed, int64_t first, int64_t R, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && R > 0 && (p || n == 0), "gen_indices: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL(gen_idx_kernel, dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), p, n, seed, first, R);
  FFH_LAUNCH_CHECK(c, "gen_indices");
  return FFH_OK;
}
int ffh_gen_uniform01(ffh_ctx* c, float* p, int64_t n, uint64_t seed, int64_t first, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && (p || n == 0), "gen_uniform01: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL((gen_f32_kernel<1>), dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), p, n, seed, first, 0.f, 1.f);
  FFH_LAUNCH_CHECK(c, "gen_uniform01");
  return FFH_OK;
}
int ffh_gen_bernoulli(ffh_ctx* c, float* p, int64_t n, uint64_t seed, int64_t first, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && (p || n == 0), "gen_bernoulli: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL((gen_f32_kernel<2>), dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), p, n, seed, first, 0.f, 1.f);
  FFH_LAUNCH_CHECK(c, "gen_bernoulli");
  return FFH_OK;
}

}  // extern "C"
