// interaction.hip -- the pairwise-dot feature interaction of DLRM as ONE kernel each way (north_star: "the pairwise-dot
// feature interaction ... MFMA-tiled").
//
// The reference spells the interaction as an operator chain -- cat -> reshape -> transpose -> batch_matmul -> flatten
// [ref: tests/ops/test_harness.py:125-177]; its driver leaves it a TODO [ref: examples/cpp/DLRM/dlrm.cc:53-54] -- and
// MLPerf-DLRM keeps only the strict lower triangle of Z Z^T.  That chain moves each sample's 27 x 128 block through HBM
// five times forward and eight times backward.  Here a WAVE owns a sample:
//   forward   Z (c <= 32 rows of d floats) goes straight from global memory into the A and B registers of
//             v_mfma_f32_32x32x2_f32 -- they are the SAME registers, because B = Z^T: lane (row r, half h) supplies
//             A(r, k) and B(k, r) = Z(r, k); which k a lane holds is free as long as A and B agree, so every lane
//             reads its half of a row as 16-byte pieces.  The 32 x 32 accumulator then holds Z Z^T; the entries i > j
//             leave as runs of consecutive floats (out[d + i (i - 1) / 2 + j]), next to a copy of row 0 (the
//             bottom-MLP output) in out[0 .. d).
//   backward  dZ = (G + G^T) Z with G the strict-lower matrix of the incoming gradient, plus the direct path of row 0.
//             S = G + G^T is assembled per wave in LDS (4.2 KB), A = S (32 x 32), B = Z in 128-column chunks whose
//             columns are permuted so that a lane loads and stores float4 (column 4 n' + t belongs to lane n' of
//             MFMA tile t); 64 MFMAs per 128 columns.
// HBM-bound: 4 (c d + d + c (c - 1) / 2) bytes per sample forward, about twice that backward; the MFMA work (64
// instructions per sample and direction at d = 128) is a few microseconds at 8192 samples.
#include "ffh_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxC = 32;           // rows of Z (1 + number of tables)

// lane (r, h) of the wave holds 16 consecutive floats of row r per 32-wide k chunk: floats [32 q + 16 h, 32 q + 16 h + 16)
template <int VEC>
__device__ __forceinline__ void load_row_piece(float (&v)[16], const float* row, int k0, int d, bool row_ok) {
#pragma unroll
  for (int u = 0; u < 16; u += VEC) {
    const int k = k0 + u;
    if (VEC == 4) {
      if (row_ok && k + 3 < d) {
        const float4 t = *reinterpret_cast<const float4*>(row + k);
        v[u] = t.x; v[u + 1] = t.y; v[u + 2] = t.z; v[u + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) v[u + e] = (row_ok && k + e < d) ? row[k + e] : 0.0f;
      }
    } else {
      v[u] = (row_ok && k < d) ? row[k] : 0.0f;
    }
  }
}

// VEC: floats per load of z (4 needs 16-byte aligned rows); OV: floats per store of the pass-through columns (the output
// row is often odd-sized -- 479 floats for MLPerf -- so its alignment is a separate matter)
template <int VEC, int OV>
__global__ __launch_bounds__(256) void dot_interaction_fwd_kernel(const float* __restrict__ z, int64_t ldz, float* __restrict__ out, int64_t ldo,
                                                                  int64_t batch, int c, int d) {
  ffh_kernel_prio();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < batch; b += nwaves) {
    const float* zb = z + b * ldz;
    const float* row = zb + (int64_t)r * d;
    const bool row_ok = r < c;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    for (int k0 = 0; k0 < d; k0 += 64) {          // two 32-wide chunks in flight
      float v0[16], v1[16];
      load_row_piece<VEC>(v0, row, k0 + 16 * h, d, row_ok);
      load_row_piece<VEC>(v1, row, k0 + 32 + 16 * h, d, row_ok);
#pragma unroll
      for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[u], v0[u], acc, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[u], v1[u], acc, 0, 0, 0);
    }
    float* ob = out + b * ldo;
    // the bottom-MLP output passes through: out[0 .. d) = Z[0][:]
    for (int k = lane * OV; k < d; k += 64 * OV) {
      if (OV == 4 && k + 3 < d) {
        *reinterpret_cast<float4*>(ob + k) = *reinterpret_cast<const float4*>(zb + k);
      } else {
        for (int e = 0; e < OV && k + e < d; e++) ob[k + e] = zb[k + e];
      }
    }
    // accumulator: lane holds column j = r, rows i = 8 (v / 4) + 4 h + v % 4
    float* tri = ob + d;
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int i = 8 * (v >> 2) + 4 * h + (v & 3);
      if (i < c && r < i) tri[i * (i - 1) / 2 + r] = acc[v];
    }
  }
}

// Forward for the shape the models use (d = 128, 16-byte aligned rows).  The register-fed kernel above reads each row as
// 16-byte pieces 64 bytes apart (32 rows x 2 pieces per instruction) because that is the MFMA operand layout: 0.40 of the
// HBM rate.  Here the sample goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = two whole 512-byte
// rows per instruction, fully coalesced, no staging registers, no ds_write), double-buffered per wave: the next sample's 14
// pieces are issued one per four MFMAs of this one.  The MFMA operand is then one conflict-free ds_read_b128 per four k
// (lane (r, h) takes k = 8 j + 4 h + e).  The image is lane-linear (a DMA writes base + 16 lane), so the swizzle that keeps the
// rows of a ds_read_b128 off each other's banks sits on the SOURCE side: position p of row r holds chunk p ^ (r & 15) (the
// lane groups of ds_read_b128 -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- hold sixteen distinct r & 15 each).  Rows >= c
// of the image are never written: whatever they hold only reaches accumulator entries with i >= c or j >= c, which are not
// stored.  8192 samples: 40.2 -> 28.5 us (0.57); 65536: 336 -> 247 us.  Measured and not better: operands read whole into
// registers with two samples in flight (29.4 us: latency is not what is left), a second accumulator chain (29.5), coalesced
// register loads + ds_write with two workgroups per CU instead of the DMA (40 us).  Issuing a 1-KiB DMA piece blocks its wave
// ~100-150 cycles; with one wave per SIMD that adds to the 64 dependent MFMAs of a sample instead of hiding under them.
typedef __attribute__((address_space(3))) void* dot_lds_ptr_t;
constexpr int kDotD = 128;                               // floats per row in this kernel
constexpr int kDotImage = kMaxC * kDotD * 4;             // 16 KB per sample image
constexpr int kDotLds = 4 * 2 * kDotImage;               // 4 waves x 2 buffers = 128 KB: one workgroup per CU

// NW = 4, NBUF = 2: one wave per SIMD, the next sample streaming into the wave's second image between this one's MFMAs (round 3).
// NW = 8, NBUF = 1 (round 5): two waves per SIMD with ONE image each -- a wave issues its sample's pieces, waits, computes; what covers
// its DMA issue stalls and its wait is the other wave's MFMAs instead of its own.
template <int OV, int NW = 4, int NBUF = 2>
__global__ __launch_bounds__(64 * NW) void dot_interaction_fwd_lds_kernel(const float* __restrict__ z, int64_t ldz, float* __restrict__ out, int64_t ldo,
                                                                      int64_t batch, int c) {
  ffh_kernel_prio();
  extern __shared__ __attribute__((aligned(16))) unsigned char dot_smem[];
  constexpr int d = kDotD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* img = dot_smem + wave * NBUF * kDotImage;
  const unsigned lds_base = (unsigned)(size_t)(dot_lds_ptr_t)img;
  const int npieces = (c + 1) / 2;                       // DMA instructions per sample: two rows each (<= 16)
  // lane-linear piece i: LDS bytes [1024 i, 1024 i + 1024) = rows 2 i (lanes 0-31) and 2 i + 1 (lanes 32-63), position p = lane & 31
  auto issue_piece = [&](const float* zb, int buf, int i) {
    int row = 2 * i + (lane >> 5);
    if (row >= c) row = c - 1;                           // the odd row behind the last one: a valid address, an unused image row
    const int p = lane & 31;
    const float* src = zb + (int64_t)row * d + 4 * (p ^ (row & 15));
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * kDotImage + i * 1024));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  int64_t b = (int64_t)blockIdx.x * NW + wave;
  int buf = 0;
  if (b < batch && NBUF == 2) {
    const float* zb = z + b * ldz;
    for (int i = 0; i < npieces; i++) issue_piece(zb, 0, i);
  }
  for (; b < batch; b += nwaves, buf ^= (NBUF - 1)) {
    if (NBUF == 1) {                                         // single image: this sample's pieces now (the image's last reader -- the previous
      const float* zb = z + b * ldz;                         // sample's stores of row 0 -- has its data in registers: wave barrier below)
      for (int i = 0; i < npieces; i++) issue_piece(zb, 0, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this sample's image is complete (its pieces were issued between the previous sample's MFMAs)
    const int64_t bn = b + nwaves;
    const bool more = NBUF == 2 && bn < batch;
    const float* zn = z + (more ? bn : b) * ldz;
    const unsigned char* im = img + buf * kDotImage + r * (d * 4);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(im + ((((2 * (j0 + u) + h) ^ (r & 15))) << 4));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].x, v[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].y, v[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].z, v[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u].w, v[u].w, acc, 0, 0, 0);
        // the next sample streams into the other buffer one 1-KiB piece per four MFMAs
        const int piece = j0 + u;
        if (more && piece < npieces) issue_piece(zn, buf ^ 1, piece);
      }
    }
    float* ob = out + b * ldo;
    // the bottom-MLP output passes through: out[0 .. d) = Z[0][:] (row 0 of the image: position p holds chunk p ^ 0)
    if (OV == 4) {
      if (lane < d / 4) *reinterpret_cast<float4*>(ob + 4 * lane) = *reinterpret_cast<const float4*>(img + buf * kDotImage + 16 * lane);
    } else {
      for (int k = lane; k < d; k += 64) ob[k] = *reinterpret_cast<const float*>(img + buf * kDotImage + 4 * k);
    }
    float* tri = ob + d;
#pragma unroll
    for (int v2 = 0; v2 < 16; v2++) {
      const int i = 8 * (v2 >> 2) + 4 * h + (v2 & 3);
      if (i < c && r < i) tri[i * (i - 1) / 2 + r] = acc[v2];
    }
    // this sample's image is overwritten by pieces issued in the NEXT iteration's MFMA loop: every ds_read of it has returned
    // by then (their results fed the MFMAs above); the wave barrier keeps the compiler from moving code across the iteration
    __builtin_amdgcn_wave_barrier();
  }
}

// backward: one wave per sample; s_S[w] is the wave's symmetric 32 x 32 (stride 33) matrix G + G^T
template <int VEC, bool ACCUM>
__global__ __launch_bounds__(256) void dot_interaction_bwd_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ og, int64_t ldg,
                                                                  float* __restrict__ zg, int64_t ldzg, int64_t batch, int c, int d) {
  ffh_kernel_prio();
  __shared__ float s_S[4][32 * 33];
  __shared__ uint16_t s_pair[kMaxC * (kMaxC - 1) / 2];     // p -> i * 33 + j
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int P = c * (c - 1) / 2;
  for (int p = threadIdx.x; p < P; p += 256) {
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (i * (i - 1) / 2 > p) i--;
    while ((i + 1) * i / 2 <= p) i++;
    s_pair[p] = (uint16_t)(i * 33 + (p - i * (i - 1) / 2));
  }
  float* S = s_S[wave];
  for (int e = lane; e < 32 * 33; e += 64) S[e] = 0.0f;      // diagonal and padding stay zero for every sample
  __syncthreads();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < batch; b += nwaves) {
    const float* gb = og + b * ldg;
    for (int p = lane; p < P; p += 64) {
      const float g = gb[d + p];
      const int ij = s_pair[p], i = ij / 33, j = ij - i * 33;
      S[ij] = g;
      S[j * 33 + i] = g;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the wave's own LDS writes have landed
    // A operand: S[r][k], lane (r, h) supplies k = 16 h + s at step s
    float a[16];
#pragma unroll
    for (int s = 0; s < 16; s++) a[s] = S[r * 33 + 16 * h + s];
    __builtin_amdgcn_wave_barrier();
    const float* zb = z + b * ldz;
    float* zgb = zg + b * ldzg;
    for (int n0 = 0; n0 < d; n0 += 128) {
      // B operand: Z[k][n0 + 4 r + t] for the four MFMA tiles t; lane (r, h) supplies row k = 16 h + s at step s
      const int col = n0 + 4 * r;
      f32x16 acc[4];
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 4) {
        float bv[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = 16 * h + s0 + u;
          const float* p = zb + (int64_t)k * d + col;
          if (VEC == 4 && k < c && col + 3 < d) {
            const float4 t4 = *reinterpret_cast<const float4*>(p);
            bv[u][0] = t4.x; bv[u][1] = t4.y; bv[u][2] = t4.z; bv[u][3] = t4.w;
          } else {
#pragma unroll
            for (int t = 0; t < 4; t++) bv[u][t] = (k < c && col + t < d) ? p[t] : 0.0f;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0 + u], bv[u][t], acc[t], 0, 0, 0);
      }
      // accumulator of tile t: lane holds column col + t, rows i = 8 (v / 4) + 4 h + v % 4
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int i = 8 * (v >> 2) + 4 * h + (v & 3);
        if (i >= c) continue;
        float o[4] = {acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
        float* dst = zgb + (int64_t)i * d + col;
        if (i == 0) {                                          // direct path of the bottom-MLP output
#pragma unroll
          for (int t = 0; t < 4; t++) if (col + t < d) o[t] = o[t] + gb[col + t];
        }
        if (VEC == 4 && col + 3 < d) {
          float4* d4 = reinterpret_cast<float4*>(dst);
          if (ACCUM) { const float4 old = *d4; o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w; }
          *d4 = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++)
            if (col + t < d) dst[t] = ACCUM ? dst[t] + o[t] : o[t];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                           // S is rewritten for the next sample
  }
}

// Backward for the shape the models use (d = 128, 16-byte aligned rows), as straight-line code.  The general kernel above guards every
// operand load and store (row < c, column < d): some 230 basic blocks, 231 registers (two waves per SIMD), a sample's operand loads
// issued behind its LDS phase -- 0.51 of the HBM rate.  Here rows past c read row c - 1 (their A columns are the zero padding of S,
// so what they hold does not matter), a pass's sixteen operand loads are issued back to back -- the first pass's before the
// gradient's triangle is spread into S, the next pass's before this pass's stores -- and NT = 2 tiles per pass keep the wave at 156
// registers (three waves per SIMD): 8192 samples 61.0 -> 46.8 us (0.64), 32768: 263 -> 193.
template <bool ACCUM, int NT>
__global__ __launch_bounds__(256) void dot_interaction_bwd_d128_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ og, int64_t ldg,
                                                                       float* __restrict__ zg, int64_t ldzg, int64_t batch, int c) {
  ffh_kernel_prio();
  static_assert(NT == 4 || NT == 2, "tiles per pass");
  constexpr int d = 128, NPASS = d / (32 * NT);
  typedef float vt __attribute__((ext_vector_type(NT)));
  __shared__ float s_S[4][32 * 33];
  __shared__ uint16_t s_pair[kMaxC * (kMaxC - 1) / 2];     // p -> i * 33 + j
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int P = c * (c - 1) / 2;
  for (int p = threadIdx.x; p < P; p += 256) {
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (i * (i - 1) / 2 > p) i--;
    while ((i + 1) * i / 2 <= p) i++;
    s_pair[p] = (uint16_t)(i * 33 + (p - i * (i - 1) / 2));
  }
  float* S = s_S[wave];
  for (int e = lane; e < 32 * 33; e += 64) S[e] = 0.0f;      // diagonal and padding stay zero for every sample
  __syncthreads();
  int krow[16];                                               // the operand rows of this lane, clamped
#pragma unroll
  for (int s = 0; s < 16; s++) { const int k = 16 * h + s; krow[s] = (k < c ? k : c - 1) * d; }
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < batch; b += nwaves) {
    const float* gb = og + b * ldg;
    const float* zb = z + b * ldz;
    float* zgb = zg + b * ldzg;
    vt bv[16];
    auto load_pass = [&](int pass) {
      const float* zp = zb + pass * 32 * NT + NT * r;
#pragma unroll
      for (int s = 0; s < 16; s++) bv[s] = *reinterpret_cast<const vt*>(zp + krow[s]);
    };
    load_pass(0);
    // the strict lower triangle of the gradient -> S = G + G^T (kMaxC (kMaxC - 1) / 2 = 496 entries: at most eight per lane)
    float gq[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int p = lane + 64 * q; gq[q] = p < P ? gb[d + p] : 0.0f; }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int p = lane + 64 * q;
      if (p < P) {
        const int ij = s_pair[p], i = ij / 33, j = ij - i * 33;
        S[ij] = gq[q];
        S[j * 33 + i] = gq[q];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the wave's own LDS writes have landed
    // A operand: S[r][k], lane (r, h) supplies k = 16 h + s at step s
    float a[16];
#pragma unroll
    for (int s = 0; s < 16; s++) a[s] = S[r * 33 + 16 * h + s];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
      const int col = pass * 32 * NT + NT * r;
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
#pragma unroll
      for (int s = 0; s < 16; s++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[s][t], acc[t], 0, 0, 0);
      if (pass + 1 < NPASS) load_pass(pass + 1);              // under this pass's stores
      // accumulator of tile t: lane holds column col + t, rows i = 8 (v / 4) + 4 h + v % 4
      vt direct;                                              // direct path of the bottom-MLP output (row 0); the gradient's rows need no alignment
#pragma unroll
      for (int t = 0; t < NT; t++) direct[t] = h == 0 ? gb[col + t] : 0.0f;
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int i = 8 * (v >> 2) + 4 * h + (v & 3);
        vt o;
#pragma unroll
        for (int t = 0; t < NT; t++) o[t] = acc[t][v];
        if (v == 0 && h == 0) o += direct;                    // i == 0 <=> v == 0, h == 0
        vt* dst = reinterpret_cast<vt*>(zgb + (int64_t)i * d + col);
        if (i < c) {
          if (ACCUM) o += *dst;
          *dst = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                           // S is rewritten for the next sample
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int ffh_dot_interaction_fwd(ffh_ctx* c, const float* z, int64_t ldz, float* out, int64_t ldo, int64_t batch, int nrows, int d, ffh_stream s) {
  FFH_REQUIRE(c, batch >= 0 && nrows >= 2 && nrows <= kMaxC && d >= 1 && ldz >= (int64_t)nrows * d &&
                     ldo >= d + (int64_t)nrows * (nrows - 1) / 2 && ((z && out) || batch == 0),
              "dot_interaction_fwd: bad args");
  if (batch == 0) return FFH_OK;
  const bool v4 = d % 4 == 0 && ldz % 4 == 0 && aligned16(z);
  const bool o4 = v4 && ldo % 4 == 0 && aligned16(out);
  const unsigned grid = ffh_grid(batch, 4, 4096);
  static const bool no_lds = FFH_LAB_INT("FFH_DOT_NO_LDS", 0) != 0;      // A/B switch
  if (v4 && d == kDotD && !no_lds) {
    // one 4-wave workgroup per CU (128 KB of LDS), every wave walks its samples with the next one streaming in
    static const bool ok4 = hipFuncSetAttribute((const void*)dot_interaction_fwd_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kDotLds) == hipSuccess;
    static const bool ok1 = hipFuncSetAttribute((const void*)dot_interaction_fwd_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kDotLds) == hipSuccess;
    static const bool ok8 = hipFuncSetAttribute((const void*)dot_interaction_fwd_lds_kernel<4, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kDotLds) == hipSuccess;
    static const int w8 = FFH_LAB_INT("FFH_DOT_W8", 0);          // A/B switch: two single-image waves per SIMD
    if (ok8 && w8 && o4) {
      unsigned g2 = (unsigned)c->num_cus;
      if ((int64_t)g2 * 8 > batch) g2 = (unsigned)((batch + 7) / 8);
      hipLaunchKernelGGL((dot_interaction_fwd_lds_kernel<4, 8, 1>), dim3(g2), dim3(512), kDotLds, as_stream(s), z, ldz, out, ldo, batch, nrows);
      FFH_LAUNCH_CHECK(c, "dot_interaction_fwd (lds, 8 waves)");
      return FFH_OK;
    }
    if (ok4 && ok1) {
      unsigned g2 = (unsigned)c->num_cus;
      if ((int64_t)g2 * 4 > batch) g2 = (unsigned)((batch + 3) / 4);
      if (o4) hipLaunchKernelGGL((dot_interaction_fwd_lds_kernel<4>), dim3(g2), dim3(256), kDotLds, as_stream(s), z, ldz, out, ldo, batch, nrows);
      else hipLaunchKernelGGL((dot_interaction_fwd_lds_kernel<1>), dim3(g2), dim3(256), kDotLds, as_stream(s), z, ldz, out, ldo, batch, nrows);
      FFH_LAUNCH_CHECK(c, "dot_interaction_fwd (lds)");
      return FFH_OK;
    }
    (void)hipGetLastError();
  }
  if (o4) hipLaunchKernelGGL((dot_interaction_fwd_kernel<4, 4>), dim3(grid), dim3(256), 0, as_stream(s), z, ldz, out, ldo, batch, nrows, d);
  else if (v4) hipLaunchKernelGGL((dot_interaction_fwd_kernel<4, 1>), dim3(grid), dim3(256), 0, as_stream(s), z, ldz, out, ldo, batch, nrows, d);
  else hipLaunchKernelGGL((dot_interaction_fwd_kernel<1, 1>), dim3(grid), dim3(256), 0, as_stream(s), z, ldz, out, ldo, batch, nrows, d);
  FFH_LAUNCH_CHECK(c, "dot_interaction_fwd");
  return FFH_OK;
}

int ffh_dot_interaction_bwd(ffh_ctx* c, const float* z, int64_t ldz, const float* out_grad, int64_t ldg, float* z_grad, int64_t ldzg,
                            int64_t batch, int nrows, int d, int flags, ffh_stream s) {
  FFH_REQUIRE(c, batch >= 0 && nrows >= 2 && nrows <= kMaxC && d >= 1 && ldz >= (int64_t)nrows * d && ldzg >= (int64_t)nrows * d &&
                     ldg >= d + (int64_t)nrows * (nrows - 1) / 2 && ((z && out_grad && z_grad) || batch == 0) &&
                     (flags & ~FFH_DOT_BWD_OVERWRITE) == 0,
              "dot_interaction_bwd: bad args");
  if (batch == 0) return FFH_OK;
  const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldzg % 4 == 0 && aligned16(z) && aligned16(z_grad);
  const bool over = (flags & FFH_DOT_BWD_OVERWRITE) != 0;
  const unsigned grid = ffh_grid(batch, 4, 4096);
  // the models' shape: the straight-line kernel (FFH_DOT_BWD_FAST: A/B in lab builds -- 0 off, 2 / 4 tiles per pass)
  static const int fast = FFH_LAB_INT("FFH_DOT_BWD_FAST", 2);
  if (fast && v4 && d == 128) {
#define FFH_DOT_BWD_F(A, N) hipLaunchKernelGGL((dot_interaction_bwd_d128_kernel<A, N>), dim3(grid), dim3(256), 0, as_stream(s), z, ldz, out_grad, ldg, z_grad, ldzg, batch, nrows)
    if (fast == 4) { if (over) FFH_DOT_BWD_F(false, 4); else FFH_DOT_BWD_F(true, 4); }
    else { if (over) FFH_DOT_BWD_F(false, 2); else FFH_DOT_BWD_F(true, 2); }
#undef FFH_DOT_BWD_F
    FFH_LAUNCH_CHECK(c, "dot_interaction_bwd (d128)");
    return FFH_OK;
  }
#define FFH_DOT_BWD(V, A) hipLaunchKernelGGL((dot_interaction_bwd_kernel<V, A>), dim3(grid), dim3(256), 0, as_stream(s), z, ldz, out_grad, ldg, z_grad, ldzg, batch, nrows, d)
  if (v4) { if (over) FFH_DOT_BWD(4, false); else FFH_DOT_BWD(4, true); }
  else { if (over) FFH_DOT_BWD(1, false); else FFH_DOT_BWD(1, true); }
#undef FFH_DOT_BWD
  FFH_LAUNCH_CHECK(c, "dot_interaction_bwd");
  return FFH_OK;
}

}  // extern "C"
