// linear.hip -- Linear forward/backward and BatchMatmul on the gfx950 matrix cores, in
// exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; a k-ordered fmaf chain,
// bit-for-bit -- no xf32/TF32 exists on gfx950), 157 TFLOP/s dense peak.
//
// One LDS-tiled kernel serves every GEMM of the path through element strides:
//   C[m][n] (op)= sum_k A(m,k) * B(n,k),   A(m,k) = A[m*sAm + k*sAk],  B(n,k) = B[n*sBn + k*sBk]
//     Linear fwd  y  = x w^T        A = x  (k-contiguous)  B = w  (k-contiguous)   + bias + activation
//     Linear dx  += dy w            A = dy (k-contiguous)  B = w  (n-contiguous)
//     Linear dw  += dy^T x          A = dy (m-contiguous)  B = x  (n-contiguous)   split-K over the batch
//     BatchMatmul fwd/bwd           the same three forms with a batch stride (grid.z)
// Tiles are staged k-major in LDS ([k][m], [k][n]) so that the MFMA operand fetch is one
// conflict-free ds_read_b32 per lane whatever the global layout; a k-contiguous global tile is
// transposed while it is written to LDS (row pad 1 -> conflict-free scalar stores), an
// m/n-contiguous one is copied with 16-B stores (row pad 4).  Double-buffered LDS, global loads
// of tile t+1 issued before the MFMAs of tile t, one barrier per k-tile.
//
// Replaces cublasSgemm x2 + cudnnActivationForward [ref: src/ops/linear.cu:436-453],
// reluBackward/sigmoid_backward + cublasSgemm x2 + cublasSgemv [ref: src/ops/linear.cu:624-659],
// cublasSgemmStridedBatched [ref: src/ops/batch_matmul.cu:238-241,393-398].
#include "ffh_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_ADD = 1, EPI_ATOMIC = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  float*       C;
  const float* bias;
  int64_t sAm, sAk, sBn, sBk, ldc;
  int64_t bsA, bsB, bsC;     // batch strides (grid.z = batch when splitk == 1)
  int M, N, K;
  int k_per_split;           // multiple of kSplitGran; grid.z = split when splitk > 1
  int splitk;
  int epi;
  int act;
};

constexpr int kSplitGran = 32;   // split-K granularity: a multiple of every BK

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == FFH_AC_MODE_RELU) return v > 0.0f ? v : 0.0f;
  if (act == FFH_AC_MODE_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// Load 4 consecutive elements along the contiguous dimension (index c0..c0+3 < climit) of
// row `r` (valid if r < rlimit).  p points at element (r, c0).
__device__ __forceinline__ float4 load4_guard(const float* p, bool row_ok, int c0, int climit, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return v;
  if (vec_ok && c0 + 3 < climit) return *reinterpret_cast<const float4*>(p);
  if (c0 + 0 < climit) v.x = p[0];
  if (c0 + 1 < climit) v.y = p[1];
  if (c0 + 2 < climit) v.z = p[2];
  if (c0 + 3 < climit) v.w = p[3];
  return v;
}

template <int BM, int BN, int BK, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int PA = AKC ? 1 : 4, PB = BKC ? 1 : 4;
  constexpr int LA = BM + PA, LB = BN + PB;
  constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;   // float4 staging slots per thread
  constexpr int KQ = BK / 4;                                // float4 per k-contiguous row
  constexpr int RPP = 256 / KQ;                             // rows per staging pass
  constexpr int WM = BM / 64, WN = BN / 64;          // 32x32 MFMA tiles per wave (2x2 waves)
  __shared__ float As[2][BK * LA];
  __shared__ float Bs[2][BK * LB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const float* A = g.A;
  const float* B = g.B;
  float* C = g.C;
  int kb = 0, ke = g.K;
  if (g.splitk > 1) {
    kb = blockIdx.z * g.k_per_split;
    ke = kb + g.k_per_split < g.K ? kb + g.k_per_split : g.K;
  } else {
    A += (int64_t)blockIdx.z * g.bsA; B += (int64_t)blockIdx.z * g.bsB; C += (int64_t)blockIdx.z * g.bsC;
  }
  if (kb >= ke) return;
  const int nk = (ke - kb + BK - 1) / BK;

  const bool a_vec = (AKC ? (g.sAm % 4 == 0) : (g.sAk % 4 == 0)) && (((uintptr_t)A & 15) == 0);
  const bool b_vec = (BKC ? (g.sBn % 4 == 0) : (g.sBk % 4 == 0)) && (((uintptr_t)B & 15) == 0);

  float4 ra[NA], rb[NB];

  auto load_tile = [&](int kt) {
    const int k0 = kb + kt * BK;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {   // rows of A are k-contiguous: 8 float4 per row
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        const int m = m0 + row, k = k0 + k4 * 4;
        ra[i] = load4_guard(A + (int64_t)m * g.sAm + k, m < g.M, k, ke, a_vec);
      } else {     // rows of the tile are k, contiguous along m
        constexpr int PER = BM / 4;
        const int m4 = tid % PER, kr = tid / PER + i * (256 / PER);
        const int m = m0 + m4 * 4, k = k0 + kr;
        ra[i] = load4_guard(A + (int64_t)k * g.sAk + m, k < ke, m, g.M, a_vec);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        const int n = n0 + row, k = k0 + k4 * 4;
        rb[i] = load4_guard(B + (int64_t)n * g.sBn + k, n < g.N, k, ke, b_vec);
      } else {
        constexpr int PER = BN / 4;
        const int n4 = tid % PER, kr = tid / PER + i * (256 / PER);
        const int n = n0 + n4 * 4, k = k0 + kr;
        rb[i] = load4_guard(B + (int64_t)k * g.sBk + n, k < ke, n, g.N, b_vec);
      }
    }
  };

  auto store_tile = [&](int buf) {
    float* as = As[buf];
    float* bs = Bs[buf];
#pragma unroll
    for (int i = 0; i < NA; i++) {
      if (AKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        as[(k4 * 4 + 0) * LA + row] = ra[i].x;
        as[(k4 * 4 + 1) * LA + row] = ra[i].y;
        as[(k4 * 4 + 2) * LA + row] = ra[i].z;
        as[(k4 * 4 + 3) * LA + row] = ra[i].w;
      } else {
        constexpr int PER = BM / 4;
        const int m4 = tid % PER, kr = tid / PER + i * (256 / PER);
        *reinterpret_cast<float4*>(&as[kr * LA + m4 * 4]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (BKC) {
        const int k4 = tid % KQ, row = tid / KQ + i * RPP;
        bs[(k4 * 4 + 0) * LB + row] = rb[i].x;
        bs[(k4 * 4 + 1) * LB + row] = rb[i].y;
        bs[(k4 * 4 + 2) * LB + row] = rb[i].z;
        bs[(k4 * 4 + 3) * LB + row] = rb[i].w;
      } else {
        constexpr int PER = BN / 4;
        const int n4 = tid % PER, kr = tid / PER + i * (256 / PER);
        *reinterpret_cast<float4*>(&bs[kr * LB + n4 * 4]) = rb[i];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
  const int lr = lane & 31, lh = lane >> 5;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; i++) a[i] = as[(2 * kk + lh) * LA + wm0 + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < WN; j++) b[j] = bs[(2 * kk + lh) * LB + wn0 + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int n = n0 + wn0 + j * 32 + lr;
      if (n >= g.N) continue;
      const float bv = (g.epi == EPI_STORE && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* cp = C + (int64_t)m * g.ldc + n;
        const float v = acc[i][j][r];
        if (g.epi == EPI_STORE) *cp = act_apply(v + bv, g.act);
        else if (g.epi == EPI_ADD) *cp = *cp + v;
        else atomicAdd(cp, v);
      }
    }
}

// dy <- dy * act'(y) in place, and db[o] += sum_b dy[b][o]; one pass over dy.
// Column sums are reduced in LDS per workgroup, then one global atomic per column.
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ db, int out, int64_t batch, int rows_per_block, int act) {
  extern __shared__ float s_col[];
  const bool want_db = db != nullptr;
  if (want_db) {
    for (int c = threadIdx.x; c < out; c += blockDim.x) s_col[c] = 0.0f;
    __syncthreads();
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < batch ? r0 + rows_per_block : batch;
  const int64_t total = (r1 - r0) * out;
  if (out >= 256 || (256 % out) != 0) {
    // general: LDS float atomics per element
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
      const int64_t r = r0 + e / out;
      const int c = (int)(e % out);
      float d = dy[r * lddy + c];
      if (act == FFH_AC_MODE_RELU) { d = (y[r * ldy + c] > 0.0f) ? d : 0.0f; dy[r * lddy + c] = d; }
      else if (act == FFH_AC_MODE_SIGMOID) { const float yo = y[r * ldy + c]; d = d * yo * (1 - yo); dy[r * lddy + c] = d; }
      if (want_db) atomicAdd(&s_col[c], d);
    }
  } else {
    // out divides 256: a thread stays on one column; sum in a register first
    const int c = threadIdx.x % out;
    float part = 0.0f;
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
      const int64_t r = r0 + e / out;
      float d = dy[r * lddy + c];
      if (act == FFH_AC_MODE_RELU) { d = (y[r * ldy + c] > 0.0f) ? d : 0.0f; dy[r * lddy + c] = d; }
      else if (act == FFH_AC_MODE_SIGMOID) { const float yo = y[r * ldy + c]; d = d * yo * (1 - yo); dy[r * lddy + c] = d; }
      part += d;
    }
    if (want_db) atomicAdd(&s_col[c], part);
  }
  if (want_db) {
    __syncthreads();
    for (int c = threadIdx.x; c < out; c += blockDim.x) atomicAdd(&db[c], s_col[c]);
  }
}

template <bool AKC, bool BKC>
int launch_gemm(ffh_ctx* c, GemmArgs& g, int64_t batch, ffh_stream s, const char* name) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || batch <= 0) return FFH_OK;
  const int64_t tiles128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * batch;
  const bool big = tiles128 >= 2 * c->num_cus && g.M >= 128 && g.N >= 128;
  const int BMv = big ? 128 : 64;
  const int gx = (g.N + BMv - 1) / BMv, gy = (g.M + BMv - 1) / BMv;
  int gz = (int)batch;
  g.splitk = 1;
  g.k_per_split = g.K;
  if (g.epi == EPI_ATOMIC) {
    // split K so that about two workgroups per CU are in flight; each split is a multiple of BK
    const int64_t tiles = (int64_t)gx * gy;
    int want = (int)((2LL * c->num_cus + tiles - 1) / tiles);
    const int max_split = (g.K + 4 * kSplitGran - 1) / (4 * kSplitGran);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int kps = (g.K + want - 1) / want;
    kps = (kps + kSplitGran - 1) / kSplitGran * kSplitGran;
    g.k_per_split = kps;
    g.splitk = (g.K + kps - 1) / kps;
    if (g.splitk <= 1) { g.splitk = 2; }       // keeps blockIdx.z meaning "split" (second split is empty)
    gz = g.splitk;
    if (batch != 1) return ffh_fail(c, FFH_ERR_BAD_ARG, "gemm: split-K with a batch");
  }
  if (gy > 65535 || gz > 65535) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "gemm: grid too large");
  dim3 grid(gx, gy, gz);
  if (big) hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 16, AKC, BKC>), grid, dim3(256), 0, as_stream(s), g);
  else hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 32, AKC, BKC>), grid, dim3(256), 0, as_stream(s), g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ffh_fail_hip(c, e, name);
  return FFH_OK;
}

bool act_ok(int act) { return act == FFH_AC_MODE_NONE || act == FFH_AC_MODE_RELU || act == FFH_AC_MODE_SIGMOID; }

}  // namespace

extern "C" {

int ffh_linear_fwd(ffh_ctx* c, const float* x, int64_t ldx, float* y, int64_t ldy, const float* w, const float* bias,
                   int in, int out, int64_t batch, int act, ffh_stream s) {
  FFH_REQUIRE(c, in > 0 && out > 0 && batch >= 0 && ldx >= in && ldy >= out, "linear_fwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x && y && w), "linear_fwd: null pointer");
  FFH_REQUIRE(c, batch < (1LL << 31), "linear_fwd: batch too large");
  if (!act_ok(act)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_fwd: activation not supported (NONE, RELU, SIGMOID)");
  if (batch == 0) return FFH_OK;
  GemmArgs g{};
  g.A = x; g.sAm = ldx; g.sAk = 1;
  g.B = w; g.sBn = in; g.sBk = 1;
  g.C = y; g.ldc = ldy; g.bias = bias;
  g.M = (int)batch; g.N = out; g.K = in;
  g.epi = EPI_STORE; g.act = act;
  return launch_gemm<true, true>(c, g, 1, s, "linear_fwd gemm");
}

int ffh_linear_bwd(ffh_ctx* c, const float* x, int64_t ldx, float* dx, int64_t lddx, const float* y, int64_t ldy,
                   float* dy, int64_t lddy, const float* w, float* dw, float* db,
                   int in, int out, int64_t batch, int act, ffh_stream s) {
  FFH_REQUIRE(c, in > 0 && out > 0 && batch >= 0 && ldx >= in && ldy >= out && lddy >= out && (!dx || lddx >= in), "linear_bwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (x && y && dy && w && dw), "linear_bwd: null pointer");
  FFH_REQUIRE(c, batch < (1LL << 31), "linear_bwd: batch too large");
  if (!act_ok(act)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "linear_bwd: activation not supported (NONE, RELU, SIGMOID)");
  if (batch == 0) return FFH_OK;
  // 1. activation gradient in place + bias gradient (one pass over dy)
  if (act != FFH_AC_MODE_NONE || db) {
    int rows = (int)((batch + 2 * c->num_cus - 1) / (2 * c->num_cus));
    if (rows < 4) rows = 4;
    const unsigned grid = (unsigned)((batch + rows - 1) / rows);
    hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(grid), dim3(256), (size_t)out * sizeof(float), as_stream(s),
                       dy, lddy, y, ldy, db, out, batch, rows, act);
    FFH_LAUNCH_CHECK(c, "act_bwd_bias_kernel");
  }
  // 2. dw[o][i] += sum_b dy[b][o] x[b][i]   (split-K over the batch, fp32 atomics)
  {
    GemmArgs g{};
    g.A = dy; g.sAm = 1; g.sAk = lddy;
    g.B = x; g.sBn = 1; g.sBk = ldx;
    g.C = dw; g.ldc = in;
    g.M = out; g.N = in; g.K = (int)batch;
    g.epi = EPI_ATOMIC; g.act = FFH_AC_MODE_NONE;
    int rc = launch_gemm<false, false>(c, g, 1, s, "linear_bwd dw gemm");
    if (rc) return rc;
  }
  // 3. dx[b][i] += sum_o dy[b][o] w[o][i]
  if (dx) {
    GemmArgs g{};
    g.A = dy; g.sAm = lddy; g.sAk = 1;
    g.B = w; g.sBn = 1; g.sBk = in;
    g.C = dx; g.ldc = lddx;
    g.M = (int)batch; g.N = in; g.K = out;
    g.epi = EPI_ADD; g.act = FFH_AC_MODE_NONE;
    int rc = launch_gemm<true, false>(c, g, 1, s, "linear_bwd dx gemm");
    if (rc) return rc;
  }
  return FFH_OK;
}

int ffh_bmm_fwd(ffh_ctx* c, float* o, const float* a, const float* b, int m, int n, int k, int64_t batch,
                int asd, int bsd, int seq, ffh_stream s) {
  FFH_REQUIRE(c, m > 0 && n > 0 && k > 0 && batch >= 0, "bmm_fwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (o && a && b), "bmm_fwd: null pointer");
  // strides from the full sizes, then seq_length truncation [ref: src/ops/batch_matmul.cu:212-236]
  const int lda = k, ldb = m, ldo = m;
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
  if (asd == 0 && seq >= 0) { FFH_REQUIRE(c, seq <= k && bsd == 1, "bmm_fwd: seq_length"); k = seq; }
  else if (asd == 1 && seq >= 0) { FFH_REQUIRE(c, seq <= n, "bmm_fwd: seq_length"); n = seq; }
  else FFH_REQUIRE(c, asd < 0 || seq < 0, "bmm_fwd: a_seq_length_dim");
  if (bsd == 0 && seq >= 0) { FFH_REQUIRE(c, seq <= m, "bmm_fwd: seq_length"); m = seq; }
  else if (bsd == 1 && seq >= 0) { FFH_REQUIRE(c, asd == 0 && k == seq, "bmm_fwd: seq_length"); }
  else FFH_REQUIRE(c, bsd < 0 || seq < 0, "bmm_fwd: b_seq_length_dim");
  if (batch == 0 || m == 0 || n == 0 || k == 0) return FFH_OK;
  GemmArgs g{};
  g.A = a; g.sAm = lda; g.sAk = 1; g.bsA = sa;
  g.B = b; g.sBn = 1; g.sBk = ldb; g.bsB = sb;
  g.C = o; g.ldc = ldo; g.bsC = so;
  g.M = n; g.N = m; g.K = k;
  g.epi = EPI_STORE; g.act = FFH_AC_MODE_NONE;
  return launch_gemm<true, false>(c, g, batch, s, "bmm_fwd gemm");
}

int ffh_bmm_bwd(ffh_ctx* c, const float* og, const float* a, float* ag, const float* b, float* bg,
                int m, int n, int k, int64_t batch, ffh_stream s) {
  FFH_REQUIRE(c, m > 0 && n > 0 && k > 0 && batch >= 0, "bmm_bwd: bad dims");
  FFH_REQUIRE(c, batch == 0 || (og && a && ag && b && bg), "bmm_bwd: null pointer");
  if (batch == 0) return FFH_OK;
  const int64_t sa = (int64_t)n * k, sb = (int64_t)k * m, so = (int64_t)n * m;
  {  // a_grad[r][q] += sum_c og[r][c] * b[q][c]
    GemmArgs g{};
    g.A = og; g.sAm = m; g.sAk = 1; g.bsA = so;
    g.B = b; g.sBn = m; g.sBk = 1; g.bsB = sb;
    g.C = ag; g.ldc = k; g.bsC = sa;
    g.M = n; g.N = k; g.K = m;
    g.epi = EPI_ADD;
    int rc = launch_gemm<true, true>(c, g, batch, s, "bmm_bwd a_grad gemm");
    if (rc) return rc;
  }
  {  // b_grad[q][cc] += sum_r a[r][q] * og[r][cc]
    GemmArgs g{};
    g.A = a; g.sAm = 1; g.sAk = k; g.bsA = sa;
    g.B = og; g.sBn = 1; g.sBk = m; g.bsB = so;
    g.C = bg; g.ldc = m; g.bsC = sb;
    g.M = k; g.N = m; g.K = n;
    g.epi = EPI_ADD;
    int rc = launch_gemm<false, false>(c, g, batch, s, "bmm_bwd b_grad gemm");
    if (rc) return rc;
  }
  return FFH_OK;
}

}  // extern "C"
