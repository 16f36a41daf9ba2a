// elementwise.hip -- Concat, MSE-loss backward, metrics, dense SGD, add-scaled for gfx950.
// All HBM-bound streaming kernels: 16-B accesses where alignment allows, one launch per
// op (the reference launches copy_with_stride / add_with_stride once per concat input,
// [ref: src/ops/concat.cu:243-248,353-357]).
#include "ffh_common.h"

namespace {

constexpr int kConcatMaxPerLaunch = 64;

struct ConcatArgs {
  float*  part[kConcatMaxPerLaunch];      // input (fwd) / input-grad (bwd) pointers
  int64_t blk[kConcatMaxPerLaunch];       // width of each part
  int64_t ld[kConcatMaxPerLaunch];        // leading dimension of each part
  int64_t off[kConcatMaxPerLaunch];       // column offset inside the concatenated row
  float*  big;                            // out (fwd) / out_grad (bwd)
  int64_t out_blk;
  int64_t num_blocks;
  int     n;
  int     overwrite;                      // bwd: store the slice instead of accumulating it
};

// grid.y = part; each workgroup streams rows of its part.  BWD adds (accumulates) into the part.
// VEC == 4: groups of four floats per row moved as one 16-byte access at 4-byte alignment (ld4u / st4u), the last group
// of a row whose width is not a multiple of four element by element -- so a 729-wide part of an 857-wide row (the dot
// interaction) is still moved 16 bytes at a time.
template <bool BWD, int VEC>
__global__ __launch_bounds__(256) void concat_kernel(const ConcatArgs a) {
  ffh_kernel_prio();
  const int p = blockIdx.y;
  float* part = a.part[p];
  const int64_t w = a.blk[p], ld = a.ld[p], off = a.off[p];
  const int64_t wv = (w + VEC - 1) / VEC;
  const int64_t total = a.num_blocks * wv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / wv;
    const int64_t e = (i - b * wv) * VEC;
    float* bp = a.big + b * a.out_blk + off + e;
    float* pp = part + b * ld + e;
    if (VEC == 4 && e + 4 <= w) {
      if (BWD) {
        const float4 g = ld4u(bp);
        if (a.overwrite) { st4u(pp, g); continue; }
        float4 x = ld4u(pp);
        x.x += g.x; x.y += g.y; x.z += g.z; x.w += g.w;
        st4u(pp, x);
      } else {
        st4u(bp, ld4u(pp));
      }
    } else {
      const int n = (int)((w - e) < VEC ? (w - e) : VEC);
      for (int q = 0; q < n; q++) {
        if (BWD) pp[q] = a.overwrite ? bp[q] : pp[q] + bp[q]; else bp[q] = pp[q];
      }
    }
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <bool BWD>
int concat_impl(ffh_ctx* c, float* big, int64_t out_blk, float* const* parts, const int64_t* in_blk,
                const int64_t* in_ld, int n, int64_t nblk, ffh_stream s, int overwrite = 0) {
  if (n < 0 || n > FFH_MAX_CONCAT_INPUTS || nblk < 0 || out_blk < 0) return ffh_fail(c, FFH_ERR_BAD_ARG, "concat: bad dims");
  if (n && (!big || !parts || !in_blk)) return ffh_fail(c, FFH_ERR_BAD_ARG, "concat: null pointer");
  int64_t off = 0;
  ConcatArgs a;
  a.big = big; a.out_blk = out_blk; a.num_blocks = nblk; a.n = 0; a.overwrite = overwrite;
  bool vec = true;   // 16-byte accesses at 4-byte alignment: no alignment or width condition left
  int64_t maxw = 0;
  auto flush = [&]() -> int {
    if (a.n == 0) return FFH_OK;
    dim3 grid(ffh_grid(nblk * ((maxw + 3) / 4), 256, 1024), a.n);
    if (vec) hipLaunchKernelGGL((concat_kernel<BWD, 4>), grid, dim3(256), 0, as_stream(s), a);
    else hipLaunchKernelGGL((concat_kernel<BWD, 1>), grid, dim3(256), 0, as_stream(s), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ffh_fail_hip(c, e, "concat_kernel");
    a.n = 0; maxw = 0; vec = true;
    return FFH_OK;
  };
  for (int i = 0; i < n; i++) {
    const int64_t ld = in_ld ? in_ld[i] : in_blk[i];
    if (in_blk[i] < 0 || ld < in_blk[i] || off + in_blk[i] > out_blk) return ffh_fail(c, FFH_ERR_BAD_ARG, "concat: widths do not fit");
    float* p = parts[i];
    const bool aliased = (p == big + off) && (ld == out_blk);   // producer already wrote in place
    if (p && !aliased && in_blk[i] > 0 && nblk > 0) {
      const bool v = in_blk[i] >= 4;   // narrower parts: element by element
      if (a.n && v != vec) { int rc = flush(); if (rc) return rc; }
      vec = v;
      a.part[a.n] = p; a.blk[a.n] = in_blk[i]; a.ld[a.n] = ld; a.off[a.n] = off;
      a.n++;
      maxw = in_blk[i] > maxw ? in_blk[i] : maxw;
      if (a.n == kConcatMaxPerLaunch) { int rc = flush(); if (rc) return rc; }
    } else if (!p && !BWD) {
      return ffh_fail(c, FFH_ERR_BAD_ARG, "concat_fwd: null input");
    }
    off += in_blk[i];
  }
  return flush();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_bwd_kernel(float* __restrict__ lg, const float* __restrict__ logit,
                                                      const float* __restrict__ label, int64_t n, float scale) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = logit[i] - label[i];
    lg[i] = __fmaf_rn(scale - 0.0f, d, 0.0f);       // scale_kernel: (b-a)*x + a with a = 0
  }
}

// per-workgroup reduction in registers/LDS, then one atomic per workgroup per counter
// lg != NULL: also writes the MSE gradient lg[b][i] = (logit - label) * scale (FFModel::backward's loss step)
__global__ __launch_bounds__(256) void metrics_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                      ffh_perf_metrics* __restrict__ perf, int64_t ns, int nc, int flags,
                                                      float* __restrict__ lg, float scale) {
  ffh_kernel_prio();
  __shared__ float s_f[3][4];
  __shared__ int   s_i[2][4];
  float mse_s = 0.f, rmse_s = 0.f, mae_s = 0.f;
  int all = 0, correct = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < ns; b += stride) {
    all += 1;
    if (lg) {
      for (int i = 0; i < nc; i++) {
        const float d = logits[b * nc + i] - labels[b * nc + i];
        lg[b * nc + i] = __fmaf_rn(scale - 0.0f, d, 0.0f);
      }
    }
    if (flags & 1) {
      if (nc == 1) { all += 1; correct += 1; }
      else {
        float max_val = 0.0f; int my = -1, tr = -1;
        for (int i = 0; i < nc; i++) {
          const float lv = logits[b * nc + i];
          if (my == -1 || lv > max_val) { max_val = lv; my = i; }
          if (labels[b * nc + i] > 0.9f) tr = i;
        }
        if (tr == my) correct += 1;
      }
    }
    if (flags & (2 | 4 | 8)) {
      float mse = 0.f, mae = 0.f;
      for (int i = 0; i < nc; i++) {
        const float diff = logits[b * nc + i] - labels[b * nc + i];
        mse = __fmaf_rn(diff, diff, mse);
        mae += fabsf(diff);
      }
      mse_s += mse; rmse_s += sqrtf(mse); mae_s += mae;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mse_s += __shfl_down(mse_s, o); rmse_s += __shfl_down(rmse_s, o); mae_s += __shfl_down(mae_s, o);
    all += __shfl_down(all, o); correct += __shfl_down(correct, o);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_f[0][wave] = mse_s; s_f[1][wave] = rmse_s; s_f[2][wave] = mae_s; s_i[0][wave] = all; s_i[1][wave] = correct; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = (s_f[0][0] + s_f[0][1]) + (s_f[0][2] + s_f[0][3]);
    const float r = (s_f[1][0] + s_f[1][1]) + (s_f[1][2] + s_f[1][3]);
    const float ab = (s_f[2][0] + s_f[2][1]) + (s_f[2][2] + s_f[2][3]);
    const int al = s_i[0][0] + s_i[0][1] + s_i[0][2] + s_i[0][3];
    const int co = s_i[1][0] + s_i[1][1] + s_i[1][2] + s_i[1][3];
    if (al) atomicAdd(&perf->train_all, al);
    if (co) atomicAdd(&perf->train_correct, co);
    if (flags & 2) atomicAdd(&perf->mse_loss, m);
    if (flags & 4) atomicAdd(&perf->rmse_loss, r);
    if (flags & 8) atomicAdd(&perf->mae_loss, ab);
  }
}

// ---------------------------------------------------------------------------
// Transpose: general <= 4-D permutation (one thread per output element, coalesced stores), and the
// case the dot interaction needs -- swap of the two innermost dims of [batch][r][c] -- through a
// padded 32x33 LDS tile so that loads AND stores are coalesced.
struct TransposeArgs {
  int64_t od[4], is_perm[4];   // output dims; input stride of the input dim that feeds output dim i
  int64_t vol;
  int nd;
};

template <bool BWD>
__global__ __launch_bounds__(256) void transpose_generic_kernel(float* __restrict__ dst, const float* __restrict__ src, const TransposeArgs a) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < a.vol; o += stride) {
    int64_t t = o, ii = 0;
#pragma unroll
    for (int i = 3; i >= 0; i--)
      if (i < a.nd) { const int64_t q = t % a.od[i]; t /= a.od[i]; ii += q * a.is_perm[i]; }
    if (BWD) dst[ii] += src[o];      // the map o -> ii is a bijection: no two threads meet
    else dst[o] = src[ii];
  }
}

// src [batch][R][C] -> dst [batch][C][R];  BWD: dst [batch][R][C] += src [batch][C][R]^T (same tile walk, roles swapped)
template <bool ACC>
__global__ __launch_bounds__(256) void transpose_last2_kernel(float* __restrict__ dst, const float* __restrict__ src, int R, int C) {
  ffh_kernel_prio();
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* sp = src + b * (int64_t)R * C;
  float* dp = dst + b * (int64_t)R * C;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = r0 + ty + k * 8, cc = c0 + tx;
    if (r < R && cc < C) tile[ty + k * 8][tx] = sp[(int64_t)r * C + cc];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cc = c0 + ty + k * 8, r = r0 + tx;
    if (r < R && cc < C) {
      float* p = dp + (int64_t)cc * R + r;
      if (ACC) *p += tile[tx][ty + k * 8]; else *p = tile[tx][ty + k * 8];
    }
  }
}

int transpose_launch(ffh_ctx* c, float* dst, const float* src, int nd, const int64_t* in_dims, const int* perm, bool bwd, ffh_stream s) {
  if (nd < 1 || nd > 4 || !in_dims || !perm) return ffh_fail(c, FFH_ERR_BAD_ARG, "transpose: ndim must be 1..4");
  bool seen[4] = {false, false, false, false};
  for (int i = 0; i < nd; i++) {
    if (perm[i] < 0 || perm[i] >= nd || seen[perm[i]] || in_dims[i] <= 0) return ffh_fail(c, FFH_ERR_BAD_ARG, "transpose: bad perm/dims");
    seen[perm[i]] = true;
  }
  int64_t is[4], vol = 1;
  for (int i = nd - 1; i >= 0; i--) { is[i] = (i == nd - 1) ? 1 : is[i + 1] * in_dims[i + 1]; vol *= in_dims[i]; }
  if (!dst || !src) return ffh_fail(c, FFH_ERR_BAD_ARG, "transpose: null pointer");
  // fast path: identity on the leading dims, swap of the two innermost
  bool last2 = nd >= 2 && perm[nd - 1] == nd - 2 && perm[nd - 2] == nd - 1;
  for (int i = 0; i < nd - 2; i++) last2 = last2 && perm[i] == i;
  if (last2) {
    int64_t batch = 1;
    for (int i = 0; i < nd - 2; i++) batch *= in_dims[i];
    const int R = (int)in_dims[nd - 2], C = (int)in_dims[nd - 1];
    if (batch <= 65535) {
      if (!bwd) {
        dim3 grid((C + 31) / 32, (R + 31) / 32, (unsigned)batch);
        hipLaunchKernelGGL((transpose_last2_kernel<false>), grid, dim3(256), 0, as_stream(s), dst, src, R, C);
      } else {   // src = out_grad [batch][C][R]; dst = in_grad [batch][R][C] += src^T
        dim3 grid((R + 31) / 32, (C + 31) / 32, (unsigned)batch);
        hipLaunchKernelGGL((transpose_last2_kernel<true>), grid, dim3(256), 0, as_stream(s), dst, src, C, R);
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return ffh_fail_hip(c, e, "transpose_last2_kernel");
      return FFH_OK;
    }
  }
  TransposeArgs a;
  a.nd = nd; a.vol = vol;
  for (int i = 0; i < 4; i++) { a.od[i] = 1; a.is_perm[i] = 0; }
  for (int i = 0; i < nd; i++) { a.od[i] = in_dims[perm[i]]; a.is_perm[i] = is[perm[i]]; }
  if (bwd) hipLaunchKernelGGL((transpose_generic_kernel<true>), dim3(ffh_grid(vol, 256)), dim3(256), 0, as_stream(s), dst, src, a);
  else hipLaunchKernelGGL((transpose_generic_kernel<false>), dim3(ffh_grid(vol, 256)), dim3(256), 0, as_stream(s), dst, src, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, "transpose_generic_kernel");
  return FFH_OK;
}

// the optimizers keep the mirrors of the weights they write (ffh_ctx_bf16_mirror_set / ffh_ctx_bf16x3_mirror_set) in the same pass, where the
// launch is the four-elements-per-lane form: the bf16 twin of elements 4 i .. 4 i + 3, or their three-plane image (element e0 + 4 i of the region).
// The same roundings as ffh_convert_f32_to_bf16 / _bf16x3 (same device helpers): the same bits as a conversion pass behind the update.
__device__ __forceinline__ void store_mirrors(const float4 v, unsigned short* twin, char* planes, int64_t e0, int64_t i) {
  if (twin) *reinterpret_cast<uint2*>(twin + 4 * i) = ffh_pack_bf16x4(v);
  if (planes) {
    uint2 p1, p2, p3;
    ffh_split_bf16x3(v, p1, p2, p3);
    char* d = planes + ffh_i32_off(e0 + 4 * i);
    *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + 64) = p2; *reinterpret_cast<uint2*>(d + 128) = p3;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ v,
                                                  int64_t n, float lr, float wd, float mom, int nesterov, int zero_grad,
                                                  unsigned short* __restrict__ twin, char* __restrict__ planes, int64_t e0) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nv = n / VEC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float wv[VEC], gv[VEC], vv[VEC];
    if (VEC == 4) {
      const float4 a = reinterpret_cast<const float4*>(w)[i], b = reinterpret_cast<const float4*>(g)[i];
      wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      if (mom > 0.f) { const float4 q = reinterpret_cast<const float4*>(v)[i]; vv[0] = q.x; vv[1] = q.y; vv[2] = q.z; vv[3] = q.w; }
    } else {
      wv[0] = w[i]; gv[0] = g[i];
      if (mom > 0.f) vv[0] = v[i];
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      float gt = __fmaf_rn(wd, wv[k], gv[k]);
      if (mom > 0.f) {
        vv[k] = __fmaf_rn(vv[k], mom, gt);
        gt = nesterov ? __fmaf_rn(mom, vv[k], gt) : vv[k];
      }
      wv[k] = __fmaf_rn(-lr, gt, wv[k]);
    }
    if (VEC == 4) {
      reinterpret_cast<float4*>(w)[i] = make_float4(wv[0], wv[1], wv[2], wv[3]);
      store_mirrors(make_float4(wv[0], wv[1], wv[2], wv[3]), twin, planes, e0, i);
      if (mom > 0.f) reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      w[i] = wv[0];
      if (mom > 0.f) v[i] = vv[0];
      if (zero_grad) g[i] = 0.f;
    }
  }
}

// adam_update [ref: src/runtime/optimizer_kernel.cu:206-226]; canonical rounding as stated in ff_hip.h
template <int VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float alpha_t, float b1, float b2, float wd, float eps, int zero_grad,
                                                   unsigned short* __restrict__ twin, char* __restrict__ planes, int64_t e0) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nv = n / VEC;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float wv[VEC], gv[VEC], mv[VEC], vv[VEC];
    if (VEC == 4) {
      const float4 a = reinterpret_cast<const float4*>(w)[i], b = reinterpret_cast<const float4*>(g)[i];
      const float4 c = reinterpret_cast<const float4*>(m)[i], d = reinterpret_cast<const float4*>(v)[i];
      wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      wv[0] = w[i]; gv[0] = g[i]; mv[0] = m[i]; vv[0] = v[i];
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      // plain IEEE operations (sqrtf and / are correctly rounded under hipcc's default
      // -fhip-fp32-correctly-rounded-divide-sqrt); no contraction beyond the fmaf written out
#pragma clang fp contract(off)
      const float gt = fmaf(wd, wv[k], gv[k]);
      const float t1 = omb1 * gt;
      mv[k] = fmaf(b1, mv[k], t1);
      const float t2 = omb2 * gt;
      const float t3 = t2 * gt;
      vv[k] = fmaf(b2, vv[k], t3);
      const float num = alpha_t * mv[k];
      const float den = sqrtf(vv[k]) + eps;
      const float step = num / den;
      wv[k] = wv[k] - step;
    }
    if (VEC == 4) {
      reinterpret_cast<float4*>(w)[i] = make_float4(wv[0], wv[1], wv[2], wv[3]);
      store_mirrors(make_float4(wv[0], wv[1], wv[2], wv[3]), twin, planes, e0, i);
      reinterpret_cast<float4*>(m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
      reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      w[i] = wv[0]; m[i] = mv[0]; v[i] = vv[0];
      if (zero_grad) g[i] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void add_scaled_kernel(float* __restrict__ d, const float* __restrict__ src, int64_t n, float scale) {
  ffh_kernel_prio();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = __fmaf_rn(src[i], scale, d[i]);
}

// dst[i] = sum over the slices, in slice order: one fixed fp32 chain per element (ffh_sum_slices_f32)
__global__ __launch_bounds__(256) void sum_slices_kernel(float* __restrict__ d, const float* __restrict__ src, int nslices, int64_t n, int64_t stride, int vec) {
  ffh_kernel_prio();
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += step * 4) {
      float4 v = *reinterpret_cast<const float4*>(src + i);
      for (int q = 1; q < nslices; q++) {
        const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)q * stride + i);
        v.x = v.x + t.x; v.y = v.y + t.y; v.z = v.z + t.z; v.w = v.w + t.w;
      }
      *reinterpret_cast<float4*>(d + i) = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
      float v = src[i];
      for (int q = 1; q < nslices; q++) v = v + src[(int64_t)q * stride + i];
      d[i] = v;
    }
  }
}

}  // namespace

namespace {
// strict lower triangle of [n][n] per sample.  The pair table p -> i*n + j is built once per workgroup in LDS; threads
// walk the kept entries of all samples in output order, so stores (and the gradient loads of the backward) are
// contiguous and a row's kept entries are read as one run.
constexpr int kTrilMaxN = 64;
template <bool BWD>
__global__ __launch_bounds__(256) void tril_kernel(float* __restrict__ tri, int64_t tri_ld, float* __restrict__ full, int64_t batch, int n) {
  ffh_kernel_prio();
  __shared__ uint16_t tab[kTrilMaxN * (kTrilMaxN - 1) / 2];
  const int P = n * (n - 1) / 2;
  for (int p = threadIdx.x; p < P; p += 256) {
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (i * (i - 1) / 2 > p) i--;
    while ((i + 1) * i / 2 <= p) i++;
    tab[p] = (uint16_t)(i * n + (p - i * (i - 1) / 2));
  }
  __syncthreads();
  const int64_t nn = (int64_t)n * n, total = batch * P;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t b = q / P;
    const int p = (int)(q - b * P);
    float* f = full + b * nn + tab[p];
    float* t = tri + b * tri_ld + p;
    if (BWD) *f = *f + *t;
    else *t = *f;
  }
}
}  // namespace

extern "C" {

int ffh_concat_fwd(ffh_ctx* c, float* out, int64_t out_blk, const float* const* ins, const int64_t* in_blk,
                   const int64_t* in_ld, int n, int64_t nblk, ffh_stream s) {
  return concat_impl<false>(c, out, out_blk, const_cast<float* const*>(reinterpret_cast<const float* const*>(ins)), in_blk, in_ld, n, nblk, s);
}
int ffh_concat_bwd(ffh_ctx* c, const float* og, int64_t out_blk, float* const* igs, const int64_t* in_blk,
                   const int64_t* in_ld, int n, int64_t nblk, ffh_stream s) {
  return concat_impl<true>(c, const_cast<float*>(og), out_blk, igs, in_blk, in_ld, n, nblk, s);
}

int ffh_concat_bwd_ex(ffh_ctx* c, const float* og, int64_t out_blk, float* const* igs, const int64_t* in_blk,
                      const int64_t* in_ld, int n, int64_t nblk, int flags, ffh_stream s) {
  FFH_REQUIRE(c, (flags & ~FFH_CONCAT_BWD_OVERWRITE) == 0, "concat_bwd_ex: unknown flags");
  return concat_impl<true>(c, const_cast<float*>(og), out_blk, igs, in_blk, in_ld, n, nblk, s, (flags & FFH_CONCAT_BWD_OVERWRITE) ? 1 : 0);
}

int ffh_transpose_fwd(ffh_ctx* c, float* out, const float* in, int nd, const int64_t* in_dims, const int* perm, ffh_stream s) {
  return transpose_launch(c, out, in, nd, in_dims, perm, false, s);
}
int ffh_transpose_bwd(ffh_ctx* c, float* in_grad, const float* out_grad, int nd, const int64_t* in_dims, const int* perm, ffh_stream s) {
  return transpose_launch(c, in_grad, out_grad, nd, in_dims, perm, true, s);
}

int ffh_tril_fwd(ffh_ctx* c, float* out, int64_t out_ld, const float* in, int64_t batch, int n, ffh_stream s) {
  FFH_REQUIRE(c, batch >= 0 && n >= 2 && n <= kTrilMaxN && out_ld >= (int64_t)n * (n - 1) / 2 && ((out && in) || batch == 0), "tril_fwd: bad args");
  if (batch == 0) return FFH_OK;
  hipLaunchKernelGGL((tril_kernel<false>), dim3(ffh_grid(batch * ((int64_t)n * (n - 1) / 2), 1024, 4096)), dim3(256), 0, as_stream(s), out, out_ld, const_cast<float*>(in), batch, n);
  FFH_LAUNCH_CHECK(c, "tril_fwd");
  return FFH_OK;
}
int ffh_tril_bwd(ffh_ctx* c, float* in_grad, const float* out_grad, int64_t grad_ld, int64_t batch, int n, ffh_stream s) {
  FFH_REQUIRE(c, batch >= 0 && n >= 2 && n <= kTrilMaxN && grad_ld >= (int64_t)n * (n - 1) / 2 && ((in_grad && out_grad) || batch == 0), "tril_bwd: bad args");
  if (batch == 0) return FFH_OK;
  hipLaunchKernelGGL((tril_kernel<true>), dim3(ffh_grid(batch * ((int64_t)n * (n - 1) / 2), 1024, 4096)), dim3(256), 0, as_stream(s), const_cast<float*>(out_grad), grad_ld, in_grad, batch, n);
  FFH_LAUNCH_CHECK(c, "tril_bwd");
  return FFH_OK;
}

int ffh_mse_bwd(ffh_ctx* c, float* lg, const float* logit, const float* label, int64_t n, float scale, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && ((lg && logit && label) || n == 0), "mse_bwd: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), lg, logit, label, n, scale);
  FFH_LAUNCH_CHECK(c, "mse_bwd_kernel");
  return FFH_OK;
}

int ffh_metrics_update(ffh_ctx* c, const float* logits, const float* labels, ffh_perf_metrics* perf,
                       int64_t ns, int nc, int flags, ffh_stream s) {
  FFH_REQUIRE(c, ns >= 0 && nc > 0 && perf && ((logits && labels) || ns == 0), "metrics_update: bad args");
  if (ns == 0) return FFH_OK;
  hipLaunchKernelGGL(metrics_kernel, dim3(ffh_grid(ns, 256, 256)), dim3(256), 0, as_stream(s), logits, labels, perf, ns, nc, flags,
                     (float*)nullptr, 0.0f);
  FFH_LAUNCH_CHECK(c, "metrics_kernel");
  return FFH_OK;
}

int ffh_mse_bwd_metrics(ffh_ctx* c, float* lg, const float* logit, const float* label, ffh_perf_metrics* perf,
                        int64_t ns, int nc, float scale, int flags, ffh_stream s) {
  FFH_REQUIRE(c, ns >= 0 && nc > 0 && perf && ((lg && logit && label) || ns == 0), "mse_bwd_metrics: bad args");
  if (ns == 0) return FFH_OK;
  hipLaunchKernelGGL(metrics_kernel, dim3(ffh_grid(ns, 256, 256)), dim3(256), 0, as_stream(s), logit, label, perf, ns, nc, flags, lg, scale);
  FFH_LAUNCH_CHECK(c, "metrics_kernel (with loss)");
  return FFH_OK;
}

int ffh_sgd_update_ex(ffh_ctx* c, float* w, float* g, float* v, int64_t n, float lr, float wd, float mom, int nesterov, int flags,
                      ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && ((w && g) || n == 0), "sgd_update: bad args");
  FFH_REQUIRE(c, !(mom > 0.f) || v, "sgd_update: momentum needs V");
  FFH_REQUIRE(c, (flags & ~FFH_OPT_ZERO_GRAD) == 0, "sgd_update: unknown flags");
  if (n == 0) return FFH_OK;
  const int zg = (flags & FFH_OPT_ZERO_GRAD) ? 1 : 0;
  const bool vec = al16(w) && al16(g) && (!(mom > 0.f) || al16(v)) && (n % 4 == 0);
  // the weights' mirrors: the bf16 twin (tensor-op mode) / the three-plane image (split mode) -- in the same pass where the launch is the vector form
  // (round 6: as a conversion launch behind the update it ran beside the next gather: 99 us in the split-mode step for 10 us of work), else by conversion
  unsigned short* tw = ffh_mirror_of(c, w, (size_t)n * 4);
  int col0 = 0;
  char* pl = const_cast<char*>(ffh_planes_of(c, w, (size_t)n * 4, &col0));
  const bool fused = vec && (!tw || ((uintptr_t)tw & 7) == 0) && (!pl || col0 % 4 == 0);
  if (vec) hipLaunchKernelGGL((sgd_kernel<4>), dim3(ffh_grid(n / 4, 256)), dim3(256), 0, as_stream(s), w, g, v, n, lr, wd, mom, nesterov, zg, fused ? tw : nullptr, fused ? pl : nullptr, (int64_t)col0);
  else hipLaunchKernelGGL((sgd_kernel<1>), dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), w, g, v, n, lr, wd, mom, nesterov, zg, (unsigned short*)nullptr, (char*)nullptr, (int64_t)0);
  FFH_LAUNCH_CHECK(c, "sgd_kernel");
  if (!fused && tw) return ffh_convert_f32_to_bf16(c, tw, w, n, s);
  if (!fused && pl) return ffh_convert_f32_to_bf16x3(c, w, 1, n, n, s);
  return FFH_OK;
}

int ffh_sgd_update(ffh_ctx* c, float* w, const float* g, float* v, int64_t n, float lr, float wd, float mom, int nesterov, ffh_stream s) {
  return ffh_sgd_update_ex(c, w, const_cast<float*>(g), v, n, lr, wd, mom, nesterov, 0, s);
}

int ffh_adam_update(ffh_ctx* c, float* w, float* g, float* m, float* v, int64_t n, float alpha_t, float b1, float b2, float wd, float eps,
                    int flags, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && ((w && g && m && v) || n == 0), "adam_update: bad args");
  FFH_REQUIRE(c, (flags & ~FFH_OPT_ZERO_GRAD) == 0, "adam_update: unknown flags");
  if (n == 0) return FFH_OK;
  const int zg = (flags & FFH_OPT_ZERO_GRAD) ? 1 : 0;
  const bool vec = al16(w) && al16(g) && al16(m) && al16(v) && (n % 4 == 0);
  unsigned short* tw = ffh_mirror_of(c, w, (size_t)n * 4);      // (the mirrors: as in ffh_sgd_update_ex)
  int col0 = 0;
  char* pl = const_cast<char*>(ffh_planes_of(c, w, (size_t)n * 4, &col0));
  const bool fused = vec && (!tw || ((uintptr_t)tw & 7) == 0) && (!pl || col0 % 4 == 0);
  if (vec) hipLaunchKernelGGL((adam_kernel<4>), dim3(ffh_grid(n / 4, 256)), dim3(256), 0, as_stream(s), w, g, m, v, n, alpha_t, b1, b2, wd, eps, zg, fused ? tw : nullptr, fused ? pl : nullptr, (int64_t)col0);
  else hipLaunchKernelGGL((adam_kernel<1>), dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), w, g, m, v, n, alpha_t, b1, b2, wd, eps, zg, (unsigned short*)nullptr, (char*)nullptr, (int64_t)0);
  FFH_LAUNCH_CHECK(c, "adam_kernel");
  if (!fused && tw) return ffh_convert_f32_to_bf16(c, tw, w, n, s);
  if (!fused && pl) return ffh_convert_f32_to_bf16x3(c, w, 1, n, n, s);
  return FFH_OK;
}

int ffh_add_scaled(ffh_ctx* c, float* d, const float* src, int64_t n, float scale, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && ((d && src) || n == 0), "add_scaled: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL(add_scaled_kernel, dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), d, src, n, scale);
  FFH_LAUNCH_CHECK(c, "add_scaled_kernel");
  return FFH_OK;
}

int ffh_sum_slices_f32(ffh_ctx* c, float* d, const float* src, int nslices, int64_t n, int64_t stride, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && nslices >= 1 && (nslices == 1 || stride >= n) && ((d && src) || n == 0), "sum_slices_f32: bad args");
  if (n == 0) return FFH_OK;
  const int vec = (n % 4 == 0) && (stride % 4 == 0) && al16(d) && al16(src);
  hipLaunchKernelGGL(sum_slices_kernel, dim3(ffh_grid(vec ? n / 4 : n, 256)), dim3(256), 0, as_stream(s), d, src, nslices, n, stride, vec);
  FFH_LAUNCH_CHECK(c, "sum_slices_kernel");
  return FFH_OK;
}

}  // extern "C"
