// embedding.hip -- Embedding forward (gather + bag-sum), the reference's dense atomic
// backward, and the fused backward + sparse SGD, hand-written for gfx950.
//
// HBM-bound integer/byte work.  Design rules applied:
//   * a table row is read by `D/4` adjacent lanes with one 16-B load each, so a
//     wave-instruction covers 64/(D/4) whole rows (512-B rows: two per instruction,
//     64-B rows: sixteen) -- full-line coalesced reads and writes;
//   * every lane-group keeps several independent row loads in flight (UNROLL);
//   * all tables of a model go in ONE launch (grid.y = table) instead of the
//     reference's one task per table;
//   * 64-bit addressing everywhere (a 200M x 256 table is 204.8 GB; the reference's
//     `int outputSize` [ref: src/ops/embedding.cu:226,229] would overflow);
//   * the backward never materialises the dense [R][D] gradient: row ids are radix-sorted
//     per table with LDS histograms and wave-ballot ranking, duplicate rows are reduced
//     in registers by the lane-group that owns the run, and each touched row is
//     read-modified-written exactly once.
#include "ffh_common.h"
#ifdef FFH_MSD_TIMING
#include <vector>
#endif

#include <type_traits>

namespace {

constexpr int kMaxChunks = 4;   // row chunks per lane: D <= 4*64*4 = 1024 (vector) / 256 (scalar)

struct EmbArgs {
  ffh_emb_table t[FFH_MAX_TABLES];
  unsigned short* out16[FFH_MAX_TABLES];   // forward, tensor-op mode: bf16 twin of t[i].io (same leading dimension) or null
  char* out3[FFH_MAX_TABLES];              // forward, split mode: the I32 image group that holds t[i].io[0] (ffh_ctx_bf16x3_mirror_set) or null ...
  int   out3c[FFH_MAX_TABLES];             // ... and that element's position in its group
  int64_t batch;
  int     ntables;
  int     L;
  int     D;
  int     aggr;
  int     nt;            // bit 0: nontemporal stores of the output rows; bit 1: nontemporal loads of the rows of tables of more than nt_rows rows (below)
  int64_t nt_rows;
};

// ---------------------------------------------------------------------------
// forward: out[b][:] = sum_j W[idx[b][j]][:]
// ---------------------------------------------------------------------------
// VEC = floats per lane per access (4: 16-B accesses; 1: any D / alignment)
template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void emb_fwd_kernel(const EmbArgs a) {
  ffh_kernel_prio();
  using vec_t = typename std::conditional<VEC == 4, float4, float>::type;
  const ffh_emb_table tb = a.t[blockIdx.y];
  unsigned short* const o16 = a.out16[blockIdx.y];
  char* const o3 = a.out3[blockIdx.y];
  const int o3c = a.out3c[blockIdx.y];
  const int D = a.D, L = a.L;
  const int nvec = D / VEC;                       // vectors per row
  const int lpr = nvec < 64 ? nvec : 64;          // lanes per row
  const int rpw = 64 / lpr;                       // rows per wave-instruction
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rsub = lane / lpr;                    // which of the wave's rows
  const int c0 = lane - rsub * lpr;               // first vector of the row for this lane
  const bool active = rsub < rpw;
  const int64_t rows_per_block = (int64_t)(blockDim.x >> 6) * rpw * UNROLL;
  const float inv = 1.0f / (float)L;
  const bool avg = a.aggr == FFH_AGGR_MODE_AVG;

  for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < a.batch; base += (int64_t)gridDim.x * rows_per_block) {
    const int64_t b0 = base + (int64_t)wave * rpw * UNROLL + rsub;
    for (int c = c0; c < nvec; c += lpr) {
      float acc[UNROLL][VEC];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[u][v] = 0.0f;
      for (int j = 0; j < L; j++) {
        int64_t row[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
          const int64_t b = b0 + (int64_t)u * rpw;
          row[u] = (active && b < a.batch) ? tb.idx[b * L + j] : -1;
        }
        vec_t val[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
          if (row[u] >= 0) {
            const vec_t* src = reinterpret_cast<const vec_t*>(tb.weight + row[u] * (int64_t)D) + c;
            if (VEC == 4 && (a.nt & 2) && tb.num_entries > a.nt_rows) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src)); val[u] = *reinterpret_cast<const vec_t*>(&t); }
            else val[u] = *src;
          }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
          if (row[u] >= 0) {
            const float* f = reinterpret_cast<const float*>(&val[u]);
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[u][v] = acc[u][v] + f[v];   // 0 + w first: (+0)+(-0) = +0 as the reference
          }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const int64_t b = b0 + (int64_t)u * rpw;
        if (active && b < a.batch) {
          vec_t o;
          float* f = reinterpret_cast<float*>(&o);
#pragma unroll
          for (int v = 0; v < VEC; v++) f[v] = avg ? acc[u][v] * inv : acc[u][v];
          if (VEC == 4 && (a.nt & 1)) { typedef float f4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(*reinterpret_cast<const f4*>(&o), reinterpret_cast<f4*>(tb.io + b * tb.ld) + c); }
          else reinterpret_cast<vec_t*>(tb.io + b * tb.ld)[c] = o;
          if (VEC == 4 && o16) {      // the twin the first top-MLP GEMM reads its operand from (ffh_ctx_bf16_mirror_set)
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const bf2 lo = {(__bf16)f[0], (__bf16)f[1]}, hi = {(__bf16)f[2], (__bf16)f[3]};
            reinterpret_cast<uint2*>(o16 + b * tb.ld)[c] = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
          }
          if (VEC == 4 && o3) {       // split mode: the three-plane image of the row piece (ffh_ctx_bf16x3_mirror_set; ld a multiple of 32)
            uint2 p1, p2, p3;
            ffh_split_bf16x3(make_float4(f[0], f[1], f[2], f[3]), p1, p2, p3);
            char* d = o3 + b * tb.ld * 6 + ffh_i32_off(o3c + 4 * c);
            *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + 64) = p2; *reinterpret_cast<uint2*>(d + 128) = p3;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// reference-parity dense backward: fp32 atomics into the full-table gradient.
// One dword per lane, lanes contiguous along the row: each atomic wave-instruction is
// 256 contiguous bytes, the shape the memory-side atomic units run at full rate.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emb_bwd_dense_kernel(const int64_t* __restrict__ idx, const float* __restrict__ g,
                                                            float* __restrict__ wg, int L, int D, int64_t batch,
                                                            int64_t gld, int avg) {
  ffh_kernel_prio();
  const int64_t total = batch * D;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / D;
    const int off = (int)(i - b * D);
    float gr = g[b * gld + off];
    if (avg) gr = gr / (float)L;
    for (int j = 0; j < L; j++) {
      const int64_t row = idx[b * L + j];
      atomicAdd(wg + row * (int64_t)D + off, gr);
    }
  }
}

// ---------------------------------------------------------------------------
// fused backward + SGD, step 1: per-table stable LSD radix sort of (row id, position)
// ---------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortMaxPerThread = 8;                        // tile = 256 * E entries, E in {1,2,4,8} chosen per call
constexpr int kMaxRadixBits = 9;
constexpr int kMaxRadix = 1 << kMaxRadixBits;

struct SortArgs {
  const int64_t* idx[FFH_MAX_TABLES];   // pass 0 source
  const uint2* src;                     // [nt][N] {row id, position}: ONE 8-byte element per entry -- a pass scatters one store per
  uint2*    dst;                        //   entry instead of two 4-byte ones into two arrays (the scattered stores are most of a pass)
  uint32_t* hist;                       // [nt][nblk][radix]
  int64_t   N;                          // entries per table (batch * L)
  int       nblk;
  int       shift;
  int       bits;
  int       pass;
  uint8_t   npass[FFH_MAX_TABLES];      // digits table t really has; later passes would be the identity and are skipped
  uint32_t* clear[2];                   // [nt][nclear[i]] dwords the pass-0 histogram kernel zeroes for the apply phase
  int       nclear[2];                  //   (level-1 meta slots, arrival counters)
  // bucket form (one stable pass on every table's TOP digit, the rest of the order made inside the apply launch, see msd_window):
  int       msd;                        // != 0: the digit of table t sits at shift_t[t] (0: its ids fit the digit -- the pass sorts it completely)
  uint8_t   shift_t[FFH_MAX_TABLES];
  uint32_t* bstart;                     // [nt][kMaxRadix + 1]: first sorted index of every bucket, [radix] = N (written by tile 0 of the scatter)
};

template <bool FIRST>
__device__ __forceinline__ uint32_t sort_load_key(const SortArgs& a, int t, int64_t i) {
  if (FIRST) return (uint32_t)a.idx[t][i];
  return a.src[(int64_t)t * a.N + i].x;
}

// histogram of the current digit per 2048-entry tile (LDS-staged bucketing)
template <bool FIRST, int E>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const SortArgs a) {
  ffh_kernel_prio();
  constexpr int kSortTile = kSortThreads * E;
  constexpr int kSortPerThread = E;
  __shared__ uint32_t s_hist[kMaxRadix];
  const int t = blockIdx.y, blk = blockIdx.x;
  if (FIRST) {     // every table has a pass 0: the apply phase finds its level-1 slots empty and its arrival counters at zero
#pragma unroll
    for (int r = 0; r < 2; r++) {
      uint32_t* z = a.clear[r] + (int64_t)t * a.nclear[r];
      for (int i = blk * kSortThreads + threadIdx.x; i < a.nclear[r]; i += gridDim.x * kSortThreads) z[i] = 0u;
    }
  }
  if (a.pass >= a.npass[t]) return;
  const int radix = 1 << a.bits;
  const uint32_t mask = radix - 1;
  const int shift = (FIRST && a.msd) ? (int)a.shift_t[t] : a.shift;
  for (int d = threadIdx.x; d < radix; d += kSortThreads) s_hist[d] = 0;
  __syncthreads();
  const int64_t tile0 = (int64_t)blk * kSortTile;
#pragma unroll
  for (int e = 0; e < kSortPerThread; e++) {
    const int64_t i = tile0 + e * kSortThreads + threadIdx.x;
    if (i < a.N) atomicAdd(&s_hist[(sort_load_key<FIRST>(a, t, i) >> shift) & mask], 1u);
  }
  __syncthreads();
  uint32_t* out = a.hist + ((int64_t)t * a.nblk + blk) * radix;
  for (int d = threadIdx.x; d < radix; d += kSortThreads) out[d] = s_hist[d];
}

// exclusive scan of the per-digit totals over digits (digit d = threadIdx.x + q*256) plus `before_d`, then the
// per-wave starting offsets: s_off[w][d] (in: count of digit d in wave w's entries) becomes the first
// destination of wave w's entries with digit d.
// NW = waves of the workgroup (4 in the tiled sort kernels, 16 in the small-batch kernel); a thread owns the digits
// threadIdx.x + q * 64 NW, q < QN = ceil(kMaxRadix / (64 NW))
template <int NW, int QN>
__device__ __forceinline__ void sort_scan_offsets(const uint32_t (&all_d)[QN], const uint32_t (&before_d)[QN], int radix,
                                                  uint32_t (*s_off)[kMaxRadix], uint32_t* s_scan, uint32_t* s_wsum) {
  constexpr int NT = NW * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
#pragma unroll
  for (int q = 0; q < QN; q++) {
    if (q * NT >= radix) break;
    uint32_t v = all_d[q];
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t n = __shfl_up(incl, o);
      if (lane >= o) incl += n;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {
      const uint32_t t = s_wsum[w2];
      if (w2 < wave) woff += t;
      total += t;
    }
    const int d = threadIdx.x + q * NT;
    if (d < radix) s_scan[d] = carry + woff + incl - v + before_d[q];
    carry += total;
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < QN; q++) {
    const int d = threadIdx.x + q * NT;
    if (d < radix) {
      uint32_t run = s_scan[d];
#pragma unroll
      for (int w2 = 0; w2 < NW; w2++) {
        const uint32_t cnt = s_off[w2][d];
        s_off[w2][d] = run;
        run += cnt;
      }
    }
  }
  __syncthreads();
}

// stable ranking of a wave's E x 64 entries, 64 at a time: the lanes holding the same digit find each other with
// `bits` ballots (a match-any), the rank inside the group is a popcount of the lower lanes, and the group's
// lowest lane advances the wave's running offset in LDS.  `out.put(dest, key, pos)` stores an entry (SortOutGlobal / SortOutLds).
struct SortOutGlobal { uint2* kp; __device__ __forceinline__ void put(uint32_t d, uint32_t k, uint32_t p) const { kp[d] = make_uint2(k, p); } };
struct SortOutLds { uint32_t* k; uint32_t* p; __device__ __forceinline__ void put(uint32_t d, uint32_t key, uint32_t pos) const { k[d] = key; p[d] = pos; } };
template <int E, class Out>
__device__ __forceinline__ void sort_rank_and_scatter(const uint32_t (&key)[E], const uint32_t (&pos)[E], const bool (&valid)[E],
                                                      int shift, int bits, uint32_t mask, uint32_t* wave_off, const Out out, const int ne = E) {
  const int lane = threadIdx.x & 63;
  // (an LDS-typed pointer: as a generic one the volatile accesses below stayed flat instructions -- and, in the bucket form's window
  //  sort, tripped a code-generation error of this compiler on the flat null check)
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  volatile lds_u32* my_off = (volatile lds_u32*)wave_off;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (e >= ne) break;                  // (uniform: rounds past the caller's live ones hold no entry)
    const uint32_t d = (key[e] >> shift) & mask;
    unsigned long long peers = __ballot(valid[e]);
    // (unrolled over the largest digit with a uniform exit: as a loop with a run-time trip count the compiler kept `peers` under an
    //  exec-mask loop, ~13 instructions per bit; straight-line it is a compare, two selects and two ands)
#pragma unroll
    for (int bit = 0; bit < kMaxRadixBits; bit++) {
      if (bit < bits) {
        const bool one = (d >> bit) & 1u;
        const unsigned long long bal = __ballot(one);
        peers &= one ? bal : ~bal;
      }
    }
    if (valid[e]) {
      const uint32_t base = my_off[d];
      const uint32_t rank = __popcll(peers & lt_mask);
      const uint32_t dest = base + rank;
      out.put(dest, key[e], pos[e]);
      if (rank == 0) my_off[d] = base + __popcll(peers);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// stable scatter.  Each wave owns 512 consecutive entries of the tile and ranks them 64 at a
// time: the lanes holding the same digit find each other with `bits` ballots (a match-any),
// the rank inside the group is a popcount of the lower lanes, and the group's lowest lane
// advances the wave's running offset in LDS.
template <bool FIRST, int E>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const SortArgs a) {
  ffh_kernel_prio();
  constexpr int kSortTile = kSortThreads * E;
  constexpr int kSortPerThread = E;
  if (a.pass >= a.npass[blockIdx.y]) return;
  __shared__ uint32_t s_off[4][kMaxRadix];
  __shared__ uint32_t s_scan[kMaxRadix];
  __shared__ uint32_t s_wsum[4];
  const int t = blockIdx.y, blk = blockIdx.x;
  const int radix = 1 << a.bits;
  const uint32_t mask = radix - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile0 = (int64_t)blk * kSortTile;
  const int shift = (FIRST && a.msd) ? (int)a.shift_t[t] : a.shift;

  for (int d = threadIdx.x; d < 4 * kMaxRadix; d += kSortThreads) (&s_off[0][0])[d] = 0;
  __syncthreads();

  uint32_t key[kSortPerThread], pos[kSortPerThread];
  bool valid[kSortPerThread];
#pragma unroll
  for (int e = 0; e < kSortPerThread; e++) {
    const int64_t i = tile0 + wave * (kSortTile / 4) + e * 64 + lane;
    valid[e] = i < a.N;
    if (FIRST) {
      key[e] = valid[e] ? (uint32_t)a.idx[t][i] : 0u;
      pos[e] = (uint32_t)i;
    } else {
      const uint2 kp = valid[e] ? a.src[(int64_t)t * a.N + i] : make_uint2(0u, 0u);
      key[e] = kp.x; pos[e] = kp.y;
    }
    if (valid[e]) atomicAdd(&s_off[wave][(key[e] >> shift) & mask], 1u);
  }
  __syncthreads();

  // global base of every digit for this tile: digits below (all tiles) + same digit, earlier tiles.
  // The column walk over the [tiles][radix] matrix is unrolled so that 8 L2 loads are in flight.
  const uint32_t* hist_t = a.hist + (int64_t)t * a.nblk * radix;
  uint32_t all_d[2] = {0, 0}, before_d[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int d = threadIdx.x + q * kSortThreads;
    if (d < radix) {
      uint32_t all = 0, before = 0;
      int b2 = 0;
      for (; b2 + 8 <= a.nblk; b2 += 8) {
        uint32_t h[8];
#pragma unroll
        for (int u = 0; u < 8; u++) h[u] = hist_t[(int64_t)(b2 + u) * radix + d];
#pragma unroll
        for (int u = 0; u < 8; u++) { all += h[u]; before += (b2 + u < blk) ? h[u] : 0u; }
      }
      for (; b2 < a.nblk; b2++) {
        const uint32_t h = hist_t[(int64_t)b2 * radix + d];
        all += h;
        before += (b2 < blk) ? h : 0u;
      }
      all_d[q] = all; before_d[q] = before;
    }
  }
  sort_scan_offsets<4, 2>(all_d, before_d, radix, s_off, s_scan, s_wsum);
  if (FIRST && a.msd && blk == 0) {      // tile 0 has nothing before it: its scan is the table's bucket starts
    uint32_t* bs = a.bstart + (int64_t)t * (kMaxRadix + 1);
    for (int d = threadIdx.x; d < radix; d += kSortThreads) bs[d] = s_scan[d];
    if (threadIdx.x == 0) bs[radix] = (uint32_t)a.N;
  }
  sort_rank_and_scatter<kSortPerThread>(key, pos, valid, shift, a.bits, mask, s_off[wave], SortOutGlobal{a.dst + (int64_t)t * a.N});
}

// ---------------------------------------------------------------------------
// fused backward + SGD, step 2: segmented reduce of the sorted list + row update
// ---------------------------------------------------------------------------
constexpr int kRedThreads = 256;
constexpr int kRedTile = 1024;                        // max sorted entries per workgroup; the call picks 128..1024
constexpr int kRedChunksPerTile = kRedTile / FFH_EMB_CHUNK;
static_assert(kRedTile % FFH_EMB_CHUNK == 0, "tile must hold whole chunks");

enum : uint32_t { kMetaNone = 0, kMetaFirst = 1, kMetaCont = 2 };

// What happens to a touched row once its gradient sum is complete (ffh_sparse_opt, include/ff_hip.h).  OPT is a template parameter
// of the kernels: 0 = plain SGD (the fused update of SURVEY 8a-4, unchanged instructions), 1 = sgd_update with weight decay /
// momentum / nesterov, 2 = adam_update -- the element arithmetic of sgd_kernel / adam_kernel (elementwise.hip), statement by
// statement, so a row hit by one gradient row ends up with the bits the dense optimizer gives that row.
struct OptP { float lr, wd, mom, b1, b2, eps, omb1, omb2; int nesterov; int64_t nt_rows; };   // nt_rows: tables of more rows have their rows read and written nontemporal (plain SGD, 16-byte form)

template <int VEC, int OPT>
__device__ __forceinline__ void apply_row(const OptP& o, float* wrow, float* s0row, float* s1row, int c, const float (&acc)[VEC], const bool nt = false) {
  if (OPT == 0) {
    if (VEC == 4 && nt) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 w = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wrow) + c);
      w.x = __fmaf_rn(-o.lr, acc[0], w.x); w.y = __fmaf_rn(-o.lr, acc[1], w.y);
      w.z = __fmaf_rn(-o.lr, acc[2], w.z); w.w = __fmaf_rn(-o.lr, acc[3], w.w);
      __builtin_nontemporal_store(w, reinterpret_cast<f4*>(wrow) + c);
    } else if (VEC == 4) {
      float4 w = reinterpret_cast<float4*>(wrow)[c];
      w.x = __fmaf_rn(-o.lr, acc[0], w.x); w.y = __fmaf_rn(-o.lr, acc[1], w.y);
      w.z = __fmaf_rn(-o.lr, acc[2], w.z); w.w = __fmaf_rn(-o.lr, acc[3], w.w);
      reinterpret_cast<float4*>(wrow)[c] = w;
    } else {
      wrow[c] = __fmaf_rn(-o.lr, acc[0], wrow[c]);
    }
    return;
  }
  float wv[VEC], av[VEC], bv[VEC];
  const bool has0 = OPT == 2 || o.mom > 0.f;
  if (VEC == 4) {
    const float4 w = reinterpret_cast<const float4*>(wrow)[c];
    wv[0] = w.x; wv[1] = w.y; wv[2] = w.z; wv[3] = w.w;
    if (has0) { const float4 a = reinterpret_cast<const float4*>(s0row)[c]; av[0] = a.x; av[1] = a.y; av[2] = a.z; av[3] = a.w; }
    if (OPT == 2) { const float4 b = reinterpret_cast<const float4*>(s1row)[c]; bv[0] = b.x; bv[1] = b.y; bv[2] = b.z; bv[3] = b.w; }
  } else {
    wv[0] = wrow[c];
    if (has0) av[0] = s0row[c];
    if (OPT == 2) bv[0] = s1row[c];
  }
#pragma unroll
  for (int k = 0; k < VEC; k++) {
    if (OPT == 1) {            // sgd_update [ref: src/runtime/optimizer_kernel.cu:23-41], as sgd_kernel spells it
      float gt = __fmaf_rn(o.wd, wv[k], acc[k]);
      if (o.mom > 0.f) {
        av[k] = __fmaf_rn(av[k], o.mom, gt);
        gt = o.nesterov ? __fmaf_rn(o.mom, av[k], gt) : av[k];
      }
      wv[k] = __fmaf_rn(-o.lr, gt, wv[k]);
    } else {                   // adam_update [ref: src/runtime/optimizer_kernel.cu:206-226], as adam_kernel spells it (lr = alpha_t)
#pragma clang fp contract(off)
      const float gt = fmaf(o.wd, wv[k], acc[k]);
      const float t1 = o.omb1 * gt;
      av[k] = fmaf(o.b1, av[k], t1);
      const float t2 = o.omb2 * gt;
      const float t3 = t2 * gt;
      bv[k] = fmaf(o.b2, bv[k], t3);
      const float num = o.lr * av[k];
      const float den = sqrtf(bv[k]) + o.eps;
      const float step = num / den;
      wv[k] = wv[k] - step;
    }
  }
  if (VEC == 4) {
    reinterpret_cast<float4*>(wrow)[c] = make_float4(wv[0], wv[1], wv[2], wv[3]);
    if (has0) reinterpret_cast<float4*>(s0row)[c] = make_float4(av[0], av[1], av[2], av[3]);
    if (OPT == 2) reinterpret_cast<float4*>(s1row)[c] = make_float4(bv[0], bv[1], bv[2], bv[3]);
  } else {
    wrow[c] = wv[0];
    if (has0) s0row[c] = av[0];
    if (OPT == 2) s1row[c] = bv[0];
  }
}

struct RedArgs {
  ffh_emb_table t[FFH_MAX_TABLES];
  const uint2* kp[2];       // sorted {row id, position} [nt][N]: table t ends in buffer parity[t]
  uint8_t   parity[FFH_MAX_TABLES];
  int       tile;           // sorted entries per workgroup: multiple of FFH_EMB_CHUNK, <= kRedTile
  float*    partial;        // level-0 partial rows [nt][2*nchunks][D] (2 slots per FFH_EMB_CHUNK block)
  uint2*    meta;           // [nt][2*nchunks] {kind, key}
  int64_t   N;
  int       nchunks;
  int       L;
  int       D;
  int       avg;
  float*    partial1;       // level-1 partial rows [nt][2*nchunks1][D]
  uint2*    meta1;          // level-1 slots [nt][2*nchunks1] (cleared by the sort phase)
  uint32_t* arrive;         // [nt][nchunks1 + 1] (cleared by the sort phase): tiles done per 1024-block, then 1024-blocks folded
  int       nchunks1;
  OptP      op;             // the row rule's parameters (op.lr = the plain update's lr)
  float*    s0[FFH_MAX_TABLES];   // OPT 1: momentum buffer V; OPT 2: first moment M -- [num_entries][D] like the table, or null
  float*    s1[FFH_MAX_TABLES];   // OPT 2: second moment V
  // bucket form (emb_sgd_reduce_kernel<.., MSD = true>): kp[parity] is ordered by the top digit only
  uint8_t   shift_t[FFH_MAX_TABLES];   // the digit's position (0: the table is completely sorted)
#ifdef FFH_MSD_TIMING
  unsigned long long* dbg;        // lab: [workgroup][8] s_memrealtime stamps
#endif
  const uint32_t* bstart;         // [nt][kMaxRadix + 1] bucket starts
  uint32_t* nextkey;              // [nt][nchunks1]: the row id behind each 1024-block (written by the block's last tile, read by its fold)
  int       radix;
};

template <int VEC>
__device__ __forceinline__ void load_grad(float (&dst)[VEC], const float* rowp, int c, float invdiv, bool avg) {
  if (VEC == 4) {
    const float4 v = reinterpret_cast<const float4*>(rowp)[c];
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  } else {
    dst[0] = rowp[c];
  }
  if (avg) {
#pragma unroll
    for (int v = 0; v < VEC; v++) dst[v] = dst[v] / invdiv;
  }
}

// Partial rows and slot records that one workgroup writes and ANOTHER reads inside the same launch (the folds in the tail of
// emb_sgd_reduce_kernel).  The eight XCDs' L2s are not coherent with each other for ordinary accesses inside a kernel, and an
// agent-scope fence pays for that with a write-back of the whole L2 (measured: 4x on the kernel, the L2 is full of the table
// rows just written).  Instead these few accesses are agent-scope relaxed atomics -- `sc1` stores (written through) and `sc1`
// loads (served behind the L2) -- ordered by completion: the writer waits for its stores (vmcnt) before it counts itself in, the
// reader loads after it has seen the count.  AGENT = false: the one-workgroup small-batch kernel, ordinary accesses.
__device__ __forceinline__ void xwg_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <bool AGENT>
__device__ __forceinline__ void xwg_store2(uint2* p, uint2 v) {
  if (AGENT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool AGENT>
__device__ __forceinline__ uint2 xwg_load2(const uint2* p) {
  if (!AGENT) return *p;
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
template <int VEC, bool AGENT>
__device__ __forceinline__ void xwg_store_row(float* rowp, int c, const float (&v)[VEC]) {
  if (VEC == 4) {
    if (AGENT) {
      xwg_store2<true>(reinterpret_cast<uint2*>(rowp) + 2 * c, make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1])));
      xwg_store2<true>(reinterpret_cast<uint2*>(rowp) + 2 * c + 1, make_uint2(__float_as_uint(v[2]), __float_as_uint(v[3])));
    } else {
      reinterpret_cast<float4*>(rowp)[c] = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    if (AGENT) __hip_atomic_store(rowp + c, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else rowp[c] = v[0];
  }
}
template <int VEC, bool AGENT>
__device__ __forceinline__ void xwg_load_row(float (&dst)[VEC], const float* rowp, int c) {
  if (VEC == 4) {
    if (AGENT) {
      const uint2 lo = xwg_load2<true>(reinterpret_cast<const uint2*>(rowp) + 2 * c), hi = xwg_load2<true>(reinterpret_cast<const uint2*>(rowp) + 2 * c + 1);
      dst[0] = __uint_as_float(lo.x); dst[1] = __uint_as_float(lo.y); dst[2] = __uint_as_float(hi.x); dst[3] = __uint_as_float(hi.y);
    } else {
      const float4 v = reinterpret_cast<const float4*>(rowp)[c];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
  } else {
    dst[0] = AGENT ? __hip_atomic_load(rowp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rowp[c];
  }
}

constexpr int kFoldStage = 1024;     // slot records of one fold staged in LDS (a 1024-block has 64; a table's 1024-blocks: 2 N / 1024)
struct RedShared {
  uint32_t key[kRedTile + 2];     // [0] = key before the tile, [1+i], [1+n] = key after
  uint32_t pos[kRedTile];
  uint16_t start[kRedTile + 1];
  uint32_t cnt[(kRedTile / 64) + 1];
  uint2    meta[2 * kRedChunksPerTile];   // 64 at FFH_EMB_CHUNK = 32
};

// one tile of one table; `partial_t` / `meta_t` are the table's level-0 slot arrays
template <int VEC, bool AGENT, int OPT>
__device__ __forceinline__ void reduce_tile_body(const ffh_emb_table& tb, const uint2* kp,
                                                 float* partial_t, uint2* meta_t, int64_t N, int nchunks, int tile, int tile_index,
                                                 int L, int D_, bool avg_, const OptP& op, float* st0, float* st1, RedShared& sh, const int tid = threadIdx.x,
                                                 const bool preloaded = false) {
  uint32_t* s_key = sh.key;
  uint32_t* s_pos = sh.pos;
  uint16_t* s_start = sh.start;
  uint32_t* s_cnt = sh.cnt;
  uint2* s_meta = sh.meta;
  struct { int64_t N; int nchunks, L, D, avg; } a = {N, nchunks, L, D_, avg_ ? 1 : 0};
  const int64_t tile0 = (int64_t)tile_index * tile;
  const int n = tile0 >= N ? 0 : (int)((N - tile0) < tile ? (N - tile0) : tile);   // a tile past the end still walks the barriers
  const int lane = tid & 63, wave = tid >> 6;

  if (!preloaded) {       // (preloaded: the caller has filled s_key[0 .. n + 1] and s_pos[0 .. n) -- the bucket form, msd_window)
    for (int i = tid; i < n; i += kRedThreads) {
      const uint2 e = kp[tile0 + i];
      s_key[1 + i] = e.x;
      s_pos[i] = a.L == 1 ? e.y : e.y / (uint32_t)a.L;      // the sample (gradient row) of the entry
    }
    if (tid == 0) {
      s_key[0] = (tile0 > 0 && tile0 < N) ? kp[tile0 - 1].x : 0xFFFFFFFFu;   // no valid key equals it when tile0 == 0 (checked below)
      s_key[1 + n] = (tile0 + n < N) ? kp[tile0 + n].x : 0xFFFFFFFFu;
    }
  }
  const int metas = 2 * (tile / FFH_EMB_CHUNK);
  if (tid < metas) s_meta[tid] = make_uint2(kMetaNone, 0);
  __syncthreads();

  // sub-run starts: chunk boundaries and changes of row id; compacted in order
  // entry handled by (wave, e, lane) = wave*256 + e*64 + lane keeps the list sorted
  const bool at_table_start = tile0 == 0;
  bool st[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int i = wave * 256 + e * 64 + lane;
    bool s = false;
    if (i < n) s = (i % FFH_EMB_CHUNK == 0) || (s_key[1 + i] != s_key[i]) || (i == 0 && at_table_start);
    st[e] = s;
    const unsigned long long bal = __ballot(s);
    if (lane == 0) s_cnt[wave * 4 + e] = __popcll(bal);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int q = 0; q < kRedTile / 64; q++) { const uint32_t c = s_cnt[q]; s_cnt[q] = run; run += c; }
    s_cnt[kRedTile / 64] = run;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int i = wave * 256 + e * 64 + lane;
    const unsigned long long bal = __ballot(st[e]);
    if (st[e]) s_start[s_cnt[wave * 4 + e] + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
  }
  const int S = (int)s_cnt[kRedTile / 64];
  if (tid == 0) s_start[S] = (uint16_t)n;
  __syncthreads();

  // lane-groups walk the sub-runs
  const int D = a.D;
  const int nvec = D / VEC;
  const int lpr = nvec < 64 ? nvec : 64;
  const int rpw = 64 / lpr;
  const int rsub = lane / lpr;
  const int c0 = lane - rsub * lpr;
  const int groups = (kRedThreads / 64) * rpw;
  const int gid = wave * rpw + rsub;
  const float Lf = (float)a.L;
  const bool avg = a.avg != 0;

  // (Several sub-runs per lane-group in flight at once were tried -- 2 and 4, with and without the registers capped for eight
  //  waves per SIMD -- and changed nothing: with every tile resident the kernel runs at the rate the memory system takes random
  //  512-B reads and read-modify-writes, ~5 TB/s of real traffic, not at a latency chain's.)
  if (rsub < rpw) {
    for (int k = gid; k < S; k += groups) {
      const int s = s_start[k], e = s_start[k + 1];
      const uint32_t key = s_key[1 + s];
      const bool head = (s_key[s] != key) || (s == 0 && at_table_start);
      const bool tail = (s_key[1 + e] != key) || (tile0 + e >= N);
      const int64_t chunk = (tile0 + s) / FFH_EMB_CHUNK;
      const int odd = (s % FFH_EMB_CHUNK) ? 1 : 0;
      const bool single = head && tail;
      if (!single && c0 == 0) s_meta[(int)(chunk - tile0 / FFH_EMB_CHUNK) * 2 + odd] = make_uint2(head ? kMetaFirst : kMetaCont, key);
      float* wrow = tb.weight + (int64_t)key * D;
      float* prow = partial_t + (chunk * 2 + odd) * D;
      for (int c = c0; c < nvec; c += lpr) {
        float acc[VEC];
        load_grad<VEC>(acc, tb.io + (int64_t)s_pos[s] * tb.ld, c, Lf, avg);
        int q = s + 1;
        // four independent row loads in flight, summed in order
        for (; q + 4 <= e; q += 4) {
          float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
          load_grad<VEC>(v0, tb.io + (int64_t)s_pos[q] * tb.ld, c, Lf, avg);
          load_grad<VEC>(v1, tb.io + (int64_t)s_pos[q + 1] * tb.ld, c, Lf, avg);
          load_grad<VEC>(v2, tb.io + (int64_t)s_pos[q + 2] * tb.ld, c, Lf, avg);
          load_grad<VEC>(v3, tb.io + (int64_t)s_pos[q + 3] * tb.ld, c, Lf, avg);
#pragma unroll
          for (int v = 0; v < VEC; v++) acc[v] = (((acc[v] + v0[v]) + v1[v]) + v2[v]) + v3[v];
        }
        for (; q < e; q++) {
          float v0[VEC];
          load_grad<VEC>(v0, tb.io + (int64_t)s_pos[q] * tb.ld, c, Lf, avg);
#pragma unroll
          for (int v = 0; v < VEC; v++) acc[v] = acc[v] + v0[v];
        }
        if (single) {
          apply_row<VEC, OPT>(op, wrow, OPT ? st0 + (int64_t)key * D : nullptr, OPT == 2 ? st1 + (int64_t)key * D : nullptr, c, acc, tb.num_entries > op.nt_rows);
        } else {
          xwg_store_row<VEC, AGENT>(prow, c, acc);
        }
      }
    }
  }
  __syncthreads();
  if (tid < metas) {
    const int64_t slot = (tile0 / FFH_EMB_CHUNK) * 2 + tid;
    if (slot < 2 * (int64_t)a.nchunks) xwg_store2<AGENT>(meta_t + slot, s_meta[tid]);
  }
}

// step 3: fold.  A row whose run crosses block boundaries left one partial per block at level k (slot 2b:
// the run enters block b from the left; slot 2b+1: the run starts inside block b and leaves it to the
// right).  One lane-group per starting slot adds the row's consecutive level-k partials left to right,
// stopping at the boundary of the enclosing level-(k+1) block (`ratio` level-k blocks; 0 = no boundary,
// last level).  A run that is now complete is applied to the table; otherwise its level-(k+1) partial
// is written with the same two-slots-per-block convention.  Chains are <= ratio steps long.
// one table; the lane-groups numbered group0, group0+ngroups, ... share the slots [slot_lo, slot_hi).
// `keys` (the table's sorted ids; level-0 input only, ratio > 0): whether a run goes on past the end of its level-(k+1) block is
// read off the sorted list instead of the next block's first slot -- the workgroup that folds one block (the last of the
// block's reduce tiles to finish, see emb_sgd_reduce_kernel) then needs nothing another block's tiles write.
template <int VEC, bool AGENT, int OPT>
__device__ __forceinline__ void fold_table_body(const ffh_emb_table& tb, const float* part, const uint2* meta, float* pout_t, uint2* mout_t,
                                                int nin, int ratio, int D, const OptP& op, float* st0, float* st1, int64_t slot_lo, int64_t slot_hi,
                                                int64_t group0, int64_t ngroups, const uint2* keys = nullptr,
                                                const uint2* staged = nullptr, int64_t staged_lo = 0, int staged_n = 0,
                                                const uint32_t* nextkey = nullptr) {
  // `staged`: an LDS copy of meta[staged_lo, staged_lo + staged_n) the caller fetched with one parallel load (the in-kernel folds:
  // a dependent memory round trip per slot and lane-group would otherwise be most of the fold)
  auto slot_meta = [&](int64_t sl) -> uint2 {
    if (sl >= staged_lo && sl < staged_lo + staged_n) return staged[sl - staged_lo];      // (staged_n = 0: nothing staged)
    return xwg_load2<AGENT>(meta + sl);
  };
  const int nvec = D / VEC;
  const int lpr = nvec < 64 ? nvec : 64;
  const int rpw = 64 / lpr;
  const int lane = threadIdx.x & 63;
  const int rsub = lane / lpr;
  const int c0 = lane - rsub * lpr;
  if (rsub >= rpw) return;
  for (int64_t slot = slot_lo + group0; slot < slot_hi; slot += ngroups) {
    const uint2 m = slot_meta(slot);
    if (m.x == kMetaNone) continue;
    const int64_t b = slot >> 1;
    const bool at_block_start = ratio > 0 && (b % ratio == 0) && ((slot & 1) == 0);
    if (!(m.x == kMetaFirst || (m.x == kMetaCont && at_block_start))) continue;   // consumed by the walk that starts left of it
    const int64_t B = ratio > 0 ? b / ratio : 0;
    int64_t bend = ratio > 0 ? (B + 1) * (int64_t)ratio : (int64_t)nin;
    if (bend > nin) bend = nin;
    // length of the walk (same for every column chunk): the lanes of the group look at lpr candidate blocks at once -- one
    // parallel load and a ballot instead of up to `ratio` dependent loads (which were most of this kernel's time on the
    // tables whose rows are hit thousands of times)
    int64_t b2 = b + 1;
    {
      const uint64_t gmask = (lpr == 64 ? ~0ull : ((1ull << lpr) - 1ull)) << (rsub * lpr);
      for (int64_t base = b + 1; base < bend; base += lpr) {
        const int64_t idx = base + c0;
        bool ok = false;
        if (idx < bend) { const uint2 m2 = slot_meta(2 * idx); ok = m2.x == kMetaCont && m2.y == m.y; }
        const uint64_t stop = ~(uint64_t)__ballot(ok) & gmask;          // lanes of this group whose block ends the run (or lies past bend)
        if (stop) { b2 = base + (__ffsll((unsigned long long)stop) - 1 - rsub * lpr); break; }
        b2 = base + lpr;
      }
      if (b2 > bend) b2 = bend;
    }
    bool cont_after = false;
    if (b2 == bend && bend < nin) {
      // the run reached the end of the block: it continues iff the entry behind the block carries the same id (sorted list);
      // equivalently the next block's first slot is a continuation of this id
      if (nextkey) cont_after = __hip_atomic_load(nextkey + B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m.y;    // (bucket form: the list in memory is not sorted)
      else if (keys) cont_after = keys[bend * FFH_EMB_CHUNK].x == m.y;
      else { const uint2 m3 = slot_meta(2 * bend); cont_after = (m3.x == kMetaCont && m3.y == m.y); }
    }
    const bool head = m.x == kMetaFirst;
    const bool complete = head && !cont_after;
    const int64_t oslot = 2 * B + (at_block_start ? 0 : 1);
    if (!complete && c0 == 0) xwg_store2<AGENT>(mout_t + oslot, make_uint2(head ? kMetaFirst : kMetaCont, m.y));
    float* wrow = tb.weight + (int64_t)m.y * D;
    float* orow = pout_t + oslot * D;
    for (int c = c0; c < nvec; c += lpr) {
      float acc[VEC];
      xwg_load_row<VEC, AGENT>(acc, part + slot * D, c);
      int64_t q = b + 1;
      for (; q + 4 <= b2; q += 4) {   // four partial rows in flight, added in order
        float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        xwg_load_row<VEC, AGENT>(v0, part + 2 * q * D, c);
        xwg_load_row<VEC, AGENT>(v1, part + 2 * (q + 1) * D, c);
        xwg_load_row<VEC, AGENT>(v2, part + 2 * (q + 2) * D, c);
        xwg_load_row<VEC, AGENT>(v3, part + 2 * (q + 3) * D, c);
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = (((acc[v] + v0[v]) + v1[v]) + v2[v]) + v3[v];
      }
      for (; q < b2; q++) {
        float v0[VEC];
        xwg_load_row<VEC, AGENT>(v0, part + 2 * q * D, c);
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = acc[v] + v0[v];
      }
      if (complete) {
        apply_row<VEC, OPT>(op, wrow, OPT ? st0 + (int64_t)m.y * D : nullptr, OPT == 2 ? st1 + (int64_t)m.y * D : nullptr, c, acc, tb.num_entries > op.nt_rows);
      } else {
        xwg_store_row<VEC, AGENT>(orow, c, acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bucket form of the fused update (round 5; calls of <= 64 K lookups per table): ONE stable pass on the top digit of the row ids
// (radix_hist_kernel + radix_scatter_kernel with msd set: the list ends up grouped by bucket, in position order inside a bucket),
// and every tile of the apply launch makes the rest of the order for ITSELF in LDS -- three launches instead of seven at the
// per-rank shape of the 8-GPU job, where the sort was six dependent launches of ~7 us with a few kilobytes of work each.
//   * A tile needs the sorted entries [tile0 - 1, tile0 + n] (its own and the row id on either side).  An entry's sorted index
//     is its bucket's start plus its rank inside the bucket, so the tile loads every bucket that overlaps that index range WHOLE
//     (the window: ~tile + two average buckets), sorts the window by row id with the stable LDS radix passes of the small-batch
//     kernel (ids relative to the window's first bucket: two or three passes), and reads its entries off the window at
//     offset (tile0 - 1) - start(first bucket).  The canonical order (FFH_EMB_CHUNK cuts of the SORTED index) is untouched: what the
//     tile adds and where its partial rows go is decided by exactly the same list as before, so the result is bit-identical.
//   * A window larger than kWinMax entries (a hot row: thousands of hits in one bucket) is cut down to exactly the entries wanted:
//     the row id and occurrence number of the entry at a given rank of a bucket are found by counting (msd_select: one pass over
//     the bucket per nine id bits), and one more pass copies the entries between the two bounds in list order (msd_collect).  Cost
//     ~ bucket size per overlapping tile; no fallback launch, no second code path on the host.
//   * The fold of a 1024-block asks whether a run goes on behind the block: the last tile of every block leaves the row id behind
//     it in `nextkey` (the list in memory is no longer sorted).
// ---------------------------------------------------------------------------
constexpr int kWinMax = 1536;                     // entries of a tile's window: tile (<= 1024) + 2 + the two edge buckets
constexpr int kWinE = kWinMax / kRedThreads;      // ... per thread
struct alignas(16) MsdShared {
  uint32_t k[kWinMax], p[kWinMax];                // the window: row id relative to its first bucket, position
  uint32_t off[4][kMaxRadix];
  uint32_t scan[kMaxRadix];
  uint32_t bs[kMaxRadix + 1];                     // the table's bucket starts
  uint32_t wsum[4];
  uint32_t w[3][4];
  uint32_t misc[8];
};

// stable LSD radix sort of the window's `cnt` entries on the low `bitsw` bits of k[]
__device__ __forceinline__ void msd_sort_window(MsdShared& ms, const uint32_t cnt, const int bitsw) {
  if (bitsw <= 0 || cnt <= 1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int np = (bitsw + kMaxRadixBits - 1) / kMaxRadixBits, rbw = (bitsw + np - 1) / np;
  const int radixw = 1 << rbw;
  const uint32_t mask = (uint32_t)radixw - 1u;
  const int span = (((int)cnt + 3) / 4 + 63) / 64 * 64;       // consecutive entries per wave
  const int ne = span / 64;                                   // <= kWinE
  uint32_t key[kWinE], pos[kWinE];
  bool valid[kWinE];
  auto fetch = [&]() {
#pragma unroll
    for (int e = 0; e < kWinE; e++) {
      const int i = wave * span + e * 64 + lane;
      valid[e] = e < ne && i < (int)cnt;
      key[e] = valid[e] ? ms.k[i] : 0u;
      pos[e] = valid[e] ? ms.p[i] : 0u;
    }
  };
  fetch();
  __syncthreads();
  for (int p = 0; p < np; p++) {
    for (int w2 = 0; w2 < 4; w2++)
      for (int d = threadIdx.x; d < radixw; d += kRedThreads) ms.off[w2][d] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kWinE; e++)
      if (valid[e]) atomicAdd(&ms.off[wave][(key[e] >> (p * rbw)) & mask], 1u);
    __syncthreads();
    uint32_t all_d[2] = {0, 0};
    const uint32_t before_d[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int d = threadIdx.x + q * kRedThreads;
      if (d < radixw) all_d[q] = ms.off[0][d] + ms.off[1][d] + ms.off[2][d] + ms.off[3][d];
    }
    sort_scan_offsets<4, 2>(all_d, before_d, radixw, ms.off, ms.scan, ms.wsum);
    sort_rank_and_scatter<kWinE>(key, pos, valid, p * rbw, rbw, mask, ms.off[wave], SortOutLds{ms.k, ms.p}, ne);
    __syncthreads();
    if (p + 1 < np) { fetch(); __syncthreads(); }
  }
}

// The usual window (whole buckets, a few dozen entries each) without a single workgroup barrier: buckets are independent sort
// domains, so every wave takes a run of whole buckets (those that start in its quarter of the window), holds its <= 256 entries in
// registers and runs the stable passes on its own 128-counter table -- seven-bit digits of the id relative to its first bucket, two
// counters per lane for the scan, the ranking of sort_rank_and_scatter.  (A count-the-smaller-ones sort was tried first: n^2 / 256
// 64-bit compares per thread cost more than the three radix passes it replaced.)  False: some wave's share is larger -- the caller
// runs the workgroup-wide passes instead.
constexpr int kWaveE = 4;
struct SortOutLdsBase {
  uint32_t* k; uint32_t* p; uint32_t kb;
  __device__ __forceinline__ void put(uint32_t d, uint32_t key, uint32_t pos) const { k[d] = key + kb; p[d] = pos; }
};
__device__ __forceinline__ bool msd_sort_waves(MsdShared& ms, const uint32_t cnt, const int d_lo, const int d_hi, const int shift, const uint32_t ws) {
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t t_lo = (uint32_t)(((unsigned long long)wave * cnt) >> 2), t_hi = (uint32_t)(((unsigned long long)(wave + 1) * cnt) >> 2);
  int n_lo = 0, n_hi = 0;                        // buckets from d_lo on that start below t_lo / t_hi
  for (int b0 = d_lo; b0 <= d_hi; b0 += 64) {
    const int b = b0 + lane;
    const uint32_t st = b <= d_hi ? ms.bs[b] - ws : 0xFFFFFFFFu;
    n_lo += __popcll(__ballot(st < t_lo));
    n_hi += __popcll(__ballot(st < t_hi));
  }
  if (wave == 3) n_hi = d_hi - d_lo + 1;
  const int fb = d_lo + n_lo, lb = d_lo + n_hi;  // this wave's buckets [fb, lb)
  const uint32_t seg0 = ms.bs[fb] - ws, seg1 = ms.bs[lb] - ws;      // (bs[d_hi + 1] - ws = cnt)
  const uint32_t m = seg1 - seg0;
  if (lane == 0) ms.wsum[wave] = m;
  __syncthreads();
  const bool ok = ms.wsum[0] <= 64u * kWaveE && ms.wsum[1] <= 64u * kWaveE && ms.wsum[2] <= 64u * kWaveE && ms.wsum[3] <= 64u * kWaveE;
  __syncthreads();
  if (!ok) return false;
  if (m > 1) {
    int bitsw = shift;
    for (uint32_t sp = (uint32_t)(lb - fb - 1); sp; sp >>= 1) bitsw++;
    const int np = (bitsw + 6) / 7, rbw = (bitsw + np - 1) / np;
    const uint32_t mask = (1u << rbw) - 1u;
    const uint32_t kb = (uint32_t)(fb - d_lo) << shift;               // ms.k holds ids relative to bucket d_lo
    const int ne = ((int)m + 63) / 64;
    volatile lds_u32* const K = (volatile lds_u32*)(ms.k + seg0);
    volatile lds_u32* const P = (volatile lds_u32*)(ms.p + seg0);
    volatile lds_u32* const H = (volatile lds_u32*)ms.off[wave];
    uint32_t key[kWaveE], pos[kWaveE];
    bool valid[kWaveE];
#pragma unroll
    for (int e = 0; e < kWaveE; e++) {
      const uint32_t i = (uint32_t)(e * 64 + lane);
      valid[e] = e < ne && i < m;
      key[e] = valid[e] ? K[i] - kb : 0u;
      pos[e] = valid[e] ? P[i] : 0u;
    }
    for (int p = 0; p < np; p++) {
      H[lane] = 0u; H[lane + 64] = 0u;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < kWaveE; e++)
        if (valid[e]) __hip_atomic_fetch_add((lds_u32*)(H + ((key[e] >> (p * rbw)) & mask)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_wave_barrier();
      const uint32_t c0 = H[2 * lane], c1 = H[2 * lane + 1], cs = c0 + c1;
      uint32_t incl = cs;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t nb = __shfl_up(incl, o);
        if (lane >= o) incl += nb;
      }
      __builtin_amdgcn_wave_barrier();
      H[2 * lane] = incl - cs; H[2 * lane + 1] = incl - cs + c0;
      __builtin_amdgcn_wave_barrier();
      sort_rank_and_scatter<kWaveE>(key, pos, valid, p * rbw, rbw, mask, ms.off[wave], SortOutLdsBase{ms.k + seg0, ms.p + seg0, kb}, ne);
      __builtin_amdgcn_wave_barrier();
      if (p + 1 < np) {
#pragma unroll
        for (int e = 0; e < kWaveE; e++) {
          const uint32_t i = (uint32_t)(e * 64 + lane);
          if (valid[e]) { key[e] = K[i] - kb; pos[e] = P[i]; }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  return true;
}

// the entry of rank r (by row id, then list order) of the bucket kp[s, e) (every id there has the top digit dbase >> shift): its row id
// and how many entries with that id precede it.  r < e - s.
__device__ __forceinline__ void msd_select(const uint2* __restrict__ kp, const int64_t s, const int64_t e, const uint32_t r, const int shift,
                                           const uint32_t dbase, MsdShared& ms, uint32_t& row, uint32_t& app) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* h = ms.off[0];
  uint32_t pv = 0, rr = r;
  int pl = 0;
  while (pl < shift) {
    const int dg = (shift - pl) < kMaxRadixBits ? (shift - pl) : kMaxRadixBits;
    const int up = shift - pl, sh2 = up - dg;
    const uint32_t dmask = (1u << dg) - 1u;
    for (int d = tid; d < kMaxRadix; d += kRedThreads) h[d] = 0;
    __syncthreads();
    for (int64_t base = s; base < e; base += kRedThreads * 8) {
      uint32_t kk[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int64_t i = base + u * kRedThreads + tid;
        kk[u] = i < e ? kp[i].x - dbase : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int64_t i = base + u * kRedThreads + tid;
        if (i < e && (kk[u] >> up) == pv) atomicAdd(&h[(kk[u] >> sh2) & dmask], 1u);      // (up < 32; ids below the bucket's digit: kk < 2^shift)
      }
    }
    __syncthreads();
    // the digit whose candidates hold rank rr: thread t owns digits 2t, 2t + 1
    const uint32_t c0 = h[2 * tid], c1 = h[2 * tid + 1], cc = c0 + c1;
    uint32_t incl = cc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t nb = __shfl_up(incl, o);
      if (lane >= o) incl += nb;
    }
    if (lane == 63) ms.wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w2 = 0; w2 < wave; w2++) woff += ms.wsum[w2];
    const uint32_t excl = woff + incl - cc;
    if (rr >= excl && rr < excl + cc) {
      const bool second = rr >= excl + c0;
      ms.misc[0] = 2u * tid + (second ? 1u : 0u);
      ms.misc[1] = rr - excl - (second ? c0 : 0u);
    }
    __syncthreads();
    pv = (pv << dg) | ms.misc[0];
    rr = ms.misc[1];
    pl += dg;
    __syncthreads();
  }
  row = dbase + pv;
  app = rr;
}

// appends, in list order, the entries of kp[s, e) from (row0, occurrence app0) on [has_lo] and before (row1, occurrence app1) [has_hi]
// to the window; `count` (uniform) = entries in the window
__device__ __forceinline__ void msd_collect(const uint2* __restrict__ kp, const int64_t s, const int64_t e, const bool has_lo, const uint32_t row0, const uint32_t app0,
                                            const bool has_hi, const uint32_t row1, const uint32_t app1, const uint32_t kbase, MsdShared& ms, uint32_t& count) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t run0 = 0, run1 = 0;
  for (int64_t base = s; base < e; base += 4 * kRedThreads) {
    uint2 v[4];
    bool val[4], m0[4], m1[4];
    uint32_t i0[4], i1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t i = base + wave * 256 + r * 64 + lane;
      val[r] = i < e;
      v[r] = val[r] ? kp[i] : make_uint2(0u, 0u);
    }
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      m0[r] = val[r] && has_lo && v[r].x == row0;
      m1[r] = val[r] && has_hi && v[r].x == row1;
      const unsigned long long b0 = __ballot(m0[r]), b1 = __ballot(m1[r]);
      i0[r] = w0 + __popcll(b0 & lt); i1[r] = w1 + __popcll(b1 & lt);
      w0 += __popcll(b0); w1 += __popcll(b1);
    }
    if (lane == 0) { ms.w[0][wave] = w0; ms.w[1][wave] = w1; }
    __syncthreads();
    uint32_t o0 = run0, o1 = run1, t0 = 0, t1 = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      const uint32_t q0 = ms.w[0][w2], q1 = ms.w[1][w2];
      if (w2 < wave) { o0 += q0; o1 += q1; }
      t0 += q0; t1 += q1;
    }
    bool take[4];
    uint32_t tp[4], wt = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool ge = !has_lo || v[r].x > row0 || (m0[r] && o0 + i0[r] >= app0);
      const bool ltb = !has_hi || v[r].x < row1 || (m1[r] && o1 + i1[r] < app1);
      take[r] = val[r] && ge && ltb;
      const unsigned long long bt = __ballot(take[r]);
      tp[r] = wt + __popcll(bt & lt);
      wt += __popcll(bt);
    }
    if (lane == 0) ms.w[2][wave] = wt;
    __syncthreads();
    uint32_t ot = count, tt = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      const uint32_t q = ms.w[2][w2];
      if (w2 < wave) ot += q;
      tt += q;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t dst = ot + tp[r];
      if (take[r] && dst < (uint32_t)kWinMax) { ms.k[dst] = v[r].x - kbase; ms.p[dst] = v[r].y; }
    }
    run0 += t0; run1 += t1; count += tt;
    __syncthreads();
  }
}

// The sorted entries [tile0 - 1, tile0 + n] of table `kp` (grouped by top digit, bucket starts bs_g) for the reduce body: thread t
// gets elements j = t + 256 r (r < 5) of the array { id before the tile, the tile's n ids, id behind it } in okey[r] and the
// position of entry j - 1 in opos[r]; the caller copies them into RedShared (which shares its memory with ms) behind a barrier.
__device__ __forceinline__ void msd_window(const uint2* __restrict__ kp, const uint32_t* __restrict__ bs_g, const int radix, const int shift,
                                           const int64_t N, const int64_t tile0, const int n, MsdShared& ms, uint32_t (&okey)[5], uint32_t (&opos)[5]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const bool have_before = tile0 > 0, have_after = tile0 + n < N;
  const uint32_t a = (uint32_t)(have_before ? tile0 - 1 : tile0), b = (uint32_t)(tile0 + n + (have_after ? 1 : 0));      // sorted indices [a, b)
  for (int d = tid; d <= radix; d += kRedThreads) ms.bs[d] = bs_g[d];
  if (tid < 2) ms.misc[tid] = 0;
  __syncthreads();
  {
    uint32_t ca = 0, cb = 0;
    for (int d = tid; d < radix; d += kRedThreads) {
      const uint32_t v = ms.bs[d];
      ca += v <= a ? 1u : 0u;
      cb += v <= b - 1u ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ca += __shfl_xor(ca, o); cb += __shfl_xor(cb, o); }
    if (lane == 0) { atomicAdd(&ms.misc[0], ca); atomicAdd(&ms.misc[1], cb); }
  }
  __syncthreads();
  const int d_lo = (int)ms.misc[0] - 1, d_hi = (int)ms.misc[1] - 1;           // the buckets holding index a and index b - 1
  const uint32_t ws = ms.bs[d_lo], we = ms.bs[d_hi + 1];
  const uint32_t kbase = (uint32_t)d_lo << shift;
  uint32_t cnt, o;
  __syncthreads();                                                            // (misc is reused below)
  const bool whole_buckets = we - ws <= (uint32_t)kWinMax;
  if (whole_buckets) {
    cnt = we - ws; o = a - ws;
    for (uint32_t i = tid; i < cnt; i += kRedThreads) {
      const uint2 v = kp[ws + i];
      ms.k[i] = v.x - kbase; ms.p[i] = v.y;
    }
  } else {
    // cut the edge buckets down to the ranks wanted
    cnt = 0; o = 0;
    const uint32_t s_lo = ws, e_lo = ms.bs[d_lo + 1], s_hi = ms.bs[d_hi], e_hi = we;
    uint32_t row0 = 0, app0 = 0, row1 = 0, app1 = 0;
    const bool has_lo = a > s_lo, has_hi = b < e_hi;
    if (has_lo) msd_select(kp, s_lo, e_lo, a - s_lo, shift, kbase, ms, row0, app0);
    if (has_hi) msd_select(kp, s_hi, e_hi, b - s_hi, shift, (uint32_t)d_hi << shift, ms, row1, app1);
    if (d_lo == d_hi) {
      msd_collect(kp, s_lo, e_lo, has_lo, row0, app0, has_hi, row1, app1, kbase, ms, cnt);
    } else {
      msd_collect(kp, s_lo, e_lo, has_lo, row0, app0, false, 0u, 0u, kbase, ms, cnt);
      if (s_hi > e_lo) msd_collect(kp, e_lo, s_hi, false, 0u, 0u, false, 0u, 0u, kbase, ms, cnt);
      msd_collect(kp, s_hi, e_hi, false, 0u, 0u, has_hi, row1, app1, kbase, ms, cnt);
    }
  }
  __syncthreads();
  int bitsw = shift;
  for (uint32_t span = (uint32_t)(d_hi - d_lo); span; span >>= 1) bitsw++;
  if (!whole_buckets || !msd_sort_waves(ms, cnt, d_lo, d_hi, shift, ws)) msd_sort_window(ms, cnt, bitsw);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int j = tid + kRedThreads * r;
    okey[r] = 0xFFFFFFFFu; opos[r] = 0u;
    if (j <= n + 1) {
      const bool real = (j > 0 || have_before) && (j <= n || have_after);
      const uint32_t wi = o + (uint32_t)j - (have_before ? 0u : 1u);
      if (real) { okey[r] = ms.k[wi] + kbase; opos[r] = ms.p[wi]; }
    }
  }
}

// step 2 + 3 in one launch.  The folds (step 3, above) used to be two more launches; now the LAST tile of a 1024-block to finish
// folds that block's 32-block partials (the classic last-arriver reduction: an arrival counter per block, with write-through
// stores / sc1 loads of the few cross-workgroup values in place of fences, see xwg_*), and the last 1024-block of a table to be
// folded folds the table's 1024-block partials.  Who folds is decided by timing, what is added to what is not: the same additions
// in the same order as the separate launches.
// Compiled for 8 waves per SIMD (64 VGPRs; the few spills sit in the fold path): a tile's time is a chain of dependent row round
// trips, so every tile of the launch should be resident at once -- at 74 registers 1,536 of the 26-table shape's 1,664 tiles are,
// and the launch takes 336 instead of 230 us.
// OPT != 0 (momentum / weight-decay SGD, Adam on the touched rows): the row rule holds up to three more rows' worth of registers;
// those instantiations are compiled for 4 waves per SIMD instead of spilling.
// MSD: the bucket form above (the list is grouped by top digit only; six workgroups per CU: the window needs 25 KB of LDS).
struct RedSmem { RedShared sh; uint2 fmeta[kFoldStage]; };
union MsdSmem { RedSmem red; MsdShared ms; };          // the window is dead once the tile's entries sit in registers
template <bool MSD> struct RedSmemOf { typedef RedSmem type; static __device__ __forceinline__ RedSmem& red(RedSmem& s) { return s; } };
template <> struct RedSmemOf<true> { typedef MsdSmem type; static __device__ __forceinline__ RedSmem& red(MsdSmem& s) { return s.red; } };
template <int VEC, int OPT, bool MSD = false>
__global__ __launch_bounds__(kRedThreads, MSD ? (OPT == 0 ? 6 : 4) : (OPT == 0 ? 8 : 4)) void emb_sgd_reduce_kernel(const RedArgs a) {
  ffh_kernel_prio();
  __shared__ typename RedSmemOf<MSD>::type smem;
  __shared__ int s_last;
  RedShared& sh = RedSmemOf<MSD>::red(smem).sh;
  uint2* const s_fmeta = RedSmemOf<MSD>::red(smem).fmeta;
  const int tix = blockIdx.y;
  const ffh_emb_table& tb = a.t[tix];
  const uint2* keys = a.kp[a.parity[tix]] + (int64_t)tix * a.N;
  float* p0 = a.partial + (int64_t)tix * 2 * a.nchunks * a.D;
  uint2* m0 = a.meta + (int64_t)tix * 2 * a.nchunks;
  float* const st0 = a.s0[tix];
  float* const st1 = a.s1[tix];
  bool preloaded = false;
#ifdef FFH_MSD_TIMING
  unsigned long long* dbg = a.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
  if (threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
#define MSD_STAMP(i) if (threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memrealtime()
#else
#define MSD_STAMP(i)
#endif
  if constexpr (MSD) {
    const int shift = a.shift_t[tix];
    const int64_t tile0m = (int64_t)blockIdx.x * a.tile;
    const int nm = tile0m >= a.N ? 0 : (int)((a.N - tile0m) < a.tile ? (a.N - tile0m) : a.tile);
    if (shift > 0 && nm > 0) {         // (shift 0: the one pass sorted the table completely)
      uint32_t okey[5], opos[5];
      msd_window(keys, a.bstart + (int64_t)tix * (kMaxRadix + 1), a.radix, shift, a.N, tile0m, nm, smem.ms, okey, opos);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 5; r++) {
        const int j = (int)threadIdx.x + kRedThreads * r;
        if (j <= nm + 1) {
          sh.key[j] = okey[r];
          if (j >= 1 && j <= nm) sh.pos[j - 1] = a.L == 1 ? opos[r] : opos[r] / (uint32_t)a.L;
        }
      }
      preloaded = true;
    }
  }
  MSD_STAMP(1);
  reduce_tile_body<VEC, true, OPT>(tb, keys, p0, m0, a.N, a.nchunks, a.tile,
                        (int)blockIdx.x, a.L, a.D, a.avg != 0, a.op, st0, st1, sh, threadIdx.x, preloaded);
  MSD_STAMP(2);

  const int nvec = a.D / VEC;
  const int lpr = nvec < 64 ? nvec : 64;
  const int rpw = 64 / lpr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t group0 = (int64_t)wave * rpw + lane / lpr;
  const int64_t ngroups = (int64_t)(kRedThreads / 64) * rpw;
  uint32_t* arrive = a.arrive + (int64_t)tix * (a.nchunks1 + 1);
  float* p1 = a.partial1 + (int64_t)tix * 2 * a.nchunks1 * a.D;
  uint2* m1 = a.meta1 + (int64_t)tix * 2 * a.nchunks1;
  constexpr int kRatio = FFH_EMB_CHUNK1 / FFH_EMB_CHUNK;

  // this tile's partials and slots are out (written through, completed), then count it in
  const int64_t tile0 = (int64_t)blockIdx.x * a.tile;
  const int64_t B1 = tile0 / FFH_EMB_CHUNK1;
  const int64_t blk_end = (B1 + 1) * FFH_EMB_CHUNK1 < a.N ? (B1 + 1) * FFH_EMB_CHUNK1 : a.N;
  const uint32_t tiles_in_block = (uint32_t)((blk_end - B1 * FFH_EMB_CHUNK1 + a.tile - 1) / a.tile);
  uint32_t* const nextkey = MSD ? a.nextkey + (int64_t)tix * a.nchunks1 : nullptr;
  if constexpr (MSD) {
    // the block's last tile: the row id behind the block, for whoever folds it
    const int64_t tend = tile0 + a.tile < a.N ? tile0 + a.tile : a.N;
    if (threadIdx.x == 0 && tend == blk_end) __hip_atomic_store(nextkey + B1, sh.key[1 + (int)(tend - tile0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  xwg_stores_done();
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&arrive[B1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tiles_in_block - 1u;
  __syncthreads();
  if (!s_last) return;
  // the slot records the fold walks over, fetched by the whole workgroup at once; nothing to fold (the usual case on the big
  // tables, whose rows are hit once): no walk at all
  auto stage = [&](const uint2* m, int64_t lo, int64_t hi) -> bool {
    const int n = (int)(hi - lo < kFoldStage ? hi - lo : kFoldStage);
    bool any = false;
    for (int i = threadIdx.x; i < n; i += kRedThreads) {
      const uint2 v = xwg_load2<true>(m + lo + i);
      s_fmeta[i] = v;
      any |= v.x != kMetaNone;
    }
    for (int64_t i = lo + kFoldStage + threadIdx.x; i < hi; i += kRedThreads) any |= xwg_load2<true>(m + i).x != kMetaNone;   // (beyond the stage: N > 512 K)
    return __syncthreads_or(any);
  };
  if (a.nchunks1 > 1) {
    const int64_t lo = 2 * B1 * kRatio;
    const int64_t hi = lo + 2 * kRatio < 2 * (int64_t)a.nchunks ? lo + 2 * kRatio : 2 * (int64_t)a.nchunks;
    if (stage(m0, lo, hi))
      fold_table_body<VEC, true, OPT>(tb, p0, m0, p1, m1, a.nchunks, kRatio, a.D, a.op, st0, st1, lo, hi, group0, ngroups, MSD ? nullptr : keys, s_fmeta, lo, (int)(hi - lo), nextkey);
    xwg_stores_done();
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&arrive[a.nchunks1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)a.nchunks1 - 1u;
    __syncthreads();
    if (!s_last) return;
    const int64_t hi1 = 2 * (int64_t)a.nchunks1;
    if (stage(m1, 0, hi1))
      fold_table_body<VEC, true, OPT>(tb, p1, m1, p1, m1, a.nchunks1, 0, a.D, a.op, st0, st1, 0, hi1, group0, ngroups, nullptr, s_fmeta, 0, (int)(hi1 < kFoldStage ? hi1 : kFoldStage));
  } else {
    const int64_t hi0 = 2 * (int64_t)a.nchunks;
    if (stage(m0, 0, hi0))
      fold_table_body<VEC, true, OPT>(tb, p0, m0, p1, m1, a.nchunks, 0, a.D, a.op, st0, st1, 0, hi0, group0, ngroups, nullptr, s_fmeta, 0, (int)hi0);
  }
}

// ---------------------------------------------------------------------------
// small batches (N = batch*bag <= 2048 per table, e.g. the 2048-sample Criteo-Kaggle step): the whole
// chain -- LDS-resident radix sort, segmented reduce, both folds -- in ONE launch, one workgroup per
// table.  At this size the ten-launch pipeline is pure launch latency (and host issue time); the
// arithmetic and its order are identical (same device bodies), so the result is bit-identical too.
// ---------------------------------------------------------------------------
constexpr int kSmallMax = 2048;
struct SmallArgs {
  ffh_emb_table t[FFH_MAX_TABLES];
  uint8_t   npass[FFH_MAX_TABLES];
  uint2*    kp;        // [nt][N] sorted {row id, position}   (workspace)
  float*    partial0;  uint2* meta0;   // level-0 slots [nt][2*nch0]
  float*    partial1;  uint2* meta1;   // level-1 slots [nt][2*nch1]
  int64_t   N;
  int nch0, nch1, rb, L, D, avg;
  int tile;            // sorted entries per reduce team (multiple of FFH_EMB_CHUNK, <= kRedTile)
  OptP op;
  float* s0[FFH_MAX_TABLES];
  float* s1[FFH_MAX_TABLES];
};

constexpr int kSmallWaves = 8;                        // threads per table = 64 x this: sort ranks kSmallMax / threads entries per thread, reduce = teams of 256
constexpr int kSmallThreads = kSmallWaves * 64;
constexpr int kSmallRedParts = kSmallThreads / kRedThreads;

struct SmallSortShared {
  uint32_t k[kSmallMax], p[kSmallMax];
  uint32_t off[kSmallWaves][kMaxRadix];
  uint32_t scan[kMaxRadix];
  uint32_t wsum[kSmallWaves];
};
union SmallShared {                                    // the sort arrays are dead once the sorted list is in global memory
  SmallSortShared sort;
  RedShared red[kSmallRedParts];
};

template <int VEC, int OPT>
__global__ __launch_bounds__(kSmallThreads) void emb_sgd_small_kernel(const SmallArgs a) {
  ffh_kernel_prio();
  constexpr int NW = kSmallWaves;
  constexpr int E = kSmallMax / kSmallThreads;         // 2 entries per thread, wave w owns [128 w, 128 w + 128)
  __shared__ SmallShared sm;
  uint32_t* s_k = sm.sort.k;
  uint32_t* s_p = sm.sort.p;
  uint32_t (*s_off)[kMaxRadix] = sm.sort.off;
  const int tix = blockIdx.x;
  const ffh_emb_table tb = a.t[tix];
  float* const st0 = a.s0[tix];
  float* const st1 = a.s1[tix];
  const int64_t N = a.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int radix = 1 << a.rb;
  const uint32_t mask = radix - 1;

  uint32_t key[E], pos[E];
  bool valid[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = wave * (kSmallMax / NW) + e * 64 + lane;
    valid[e] = i < N;
    key[e] = valid[e] ? (uint32_t)tb.idx[i] : 0u;
    pos[e] = (uint32_t)i;
  }
  const int npass = a.npass[tix];
  for (int p = 0; p < npass; p++) {
    const int shift = p * a.rb;
    for (int d = threadIdx.x; d < NW * kMaxRadix; d += kSmallThreads) (&s_off[0][0])[d] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++)
      if (valid[e]) atomicAdd(&s_off[wave][(key[e] >> shift) & mask], 1u);
    __syncthreads();
    uint32_t all_d[1] = {0};
    const uint32_t before_d[1] = {0};
    if ((int)threadIdx.x < radix) {
      uint32_t t = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; w2++) t += s_off[w2][threadIdx.x];
      all_d[0] = t;
    }
    sort_scan_offsets<NW, 1>(all_d, before_d, radix, s_off, sm.sort.scan, sm.sort.wsum);
    sort_rank_and_scatter<E>(key, pos, valid, shift, a.rb, mask, s_off[wave], SortOutLds{s_k, s_p});
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = wave * (kSmallMax / NW) + e * 64 + lane;
      if (valid[e]) { key[e] = s_k[i]; pos[e] = s_p[i]; }
    }
    __syncthreads();
  }
  uint2* keys = a.kp + (int64_t)tix * N;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = wave * (kSmallMax / NW) + e * 64 + lane;
    if (valid[e]) keys[i] = make_uint2(key[e], pos[e]);
  }
  uint2* m1 = a.meta1 + (int64_t)tix * 2 * a.nch1;
  for (int i = threadIdx.x; i < 2 * a.nch1; i += kSmallThreads) m1[i] = make_uint2(kMetaNone, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // reduce: the 1024 threads act as four 256-thread teams, one tile of a.tile sorted entries each
  float* p0 = a.partial0 + (int64_t)tix * 2 * a.nch0 * a.D;
  uint2* m0 = a.meta0 + (int64_t)tix * 2 * a.nch0;
  const int team = threadIdx.x / kRedThreads, ttid = threadIdx.x % kRedThreads;
  const int ntiles = (int)((N + a.tile - 1) / a.tile);
  for (int t0 = 0; t0 < ntiles; t0 += kSmallRedParts) {
    reduce_tile_body<VEC, false, OPT>(tb, keys, p0, m0, N, a.nch0, a.tile, t0 + team, a.L, a.D, a.avg != 0, a.op, st0, st1, sm.red[team], ttid);
    __syncthreads();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  const int nvec = a.D / VEC;
  const int lpr = nvec < 64 ? nvec : 64;
  const int rpw = 64 / lpr;
  const int64_t group0 = (int64_t)wave * rpw + lane / lpr;
  const int64_t ngroups = (int64_t)NW * rpw;
  float* p1 = a.partial1 + (int64_t)tix * 2 * a.nch1 * a.D;
  if (a.nch1 > 1) {
    fold_table_body<VEC, false, OPT>(tb, p0, m0, p1, m1, a.nch0, FFH_EMB_CHUNK1 / FFH_EMB_CHUNK, a.D, a.op, st0, st1, 0, 2 * (int64_t)a.nch0, group0, ngroups);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    fold_table_body<VEC, false, OPT>(tb, p1, m1, p1, m1, a.nch1, 0, a.D, a.op, st0, st1, 0, 2 * (int64_t)a.nch1, group0, ngroups);
  } else {
    fold_table_body<VEC, false, OPT>(tb, p0, m0, p1, m1, a.nch0, 0, a.D, a.op, st0, st1, 0, 2 * (int64_t)a.nch0, group0, ngroups);
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct BwdLayout {
  size_t kp_a, kp_b, hist, partial, meta, partial1, meta1, arrive, bstart, total;
  int nblk, nchunks, nchunks1;
};

// entries per thread of the sort kernels: enough workgroups to cover the chip, tiles as large as that allows
inline int sort_per_thread(int nt, int64_t N) {
  const int64_t want = N * nt / (512LL * kSortThreads);
  int e = 1;
  while (e * 2 <= want && e < kSortMaxPerThread) e *= 2;
  // every scatter workgroup walks the table's [tiles][radix] histogram matrix: keep <= 32 tiles per table
  while ((N + (int64_t)kSortThreads * e - 1) / ((int64_t)kSortThreads * e) > 32 && e < kSortMaxPerThread) e *= 2;
  return e;
}
inline int reduce_tile(int nt, int64_t N) {
  const int64_t want = N * nt / 1024;
  int t = 128;
  while (t * 2 <= want && t < kRedTile) t *= 2;
  return t;
}

inline BwdLayout bwd_layout(int nt, int L, int D, int64_t batch) {
  BwdLayout l;
  const int64_t N = batch * L;
  const int sort_tile = kSortThreads * sort_per_thread(nt, N);
  l.nblk = (int)((N + sort_tile - 1) / sort_tile);
  l.nchunks = (int)((N + FFH_EMB_CHUNK - 1) / FFH_EMB_CHUNK);
  const size_t arr = align_up((size_t)nt * (size_t)N * sizeof(uint2), 256);
  size_t o = 0;
  l.kp_a = o; o += arr;
  l.kp_b = o; o += arr;
  l.hist = o; o += align_up((size_t)nt * (size_t)l.nblk * kMaxRadix * sizeof(uint32_t), 256);
  l.partial = o; o += align_up((size_t)nt * 2 * (size_t)l.nchunks * (size_t)D * sizeof(float), 256);
  l.meta = o; o += align_up((size_t)nt * 2 * (size_t)l.nchunks * sizeof(uint2), 256);
  l.nchunks1 = (int)((N + FFH_EMB_CHUNK1 - 1) / FFH_EMB_CHUNK1);
  l.partial1 = o; o += align_up((size_t)nt * 2 * (size_t)l.nchunks1 * (size_t)D * sizeof(float), 256);
  l.meta1 = o; o += align_up((size_t)nt * 2 * (size_t)l.nchunks1 * sizeof(uint2), 256);
  l.arrive = o; o += align_up((size_t)nt * ((size_t)l.nchunks1 + 1) * sizeof(uint32_t), 256);
  l.bstart = o; o += align_up((size_t)nt * (kMaxRadix + 1) * sizeof(uint32_t), 256);      // (bucket form; its nextkey array lives in `hist`, free by then)
  l.total = o;
  return l;
}

int validate_tables(ffh_ctx* c, const ffh_emb_table* t, int nt, int L, int D, int64_t batch, int aggr, const char* who) {
  if (nt < 0 || nt > FFH_MAX_TABLES) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding: ntables out of range");
  if (L <= 0 || D <= 0 || batch < 0) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding: bad dims");
  if (aggr != FFH_AGGR_MODE_SUM && aggr != FFH_AGGR_MODE_AVG) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding: aggr must be SUM or AVG");
  for (int i = 0; i < nt; i++) {
    if (batch > 0 && (!t[i].idx || !t[i].weight || !t[i].io)) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding: null pointer");
    if (t[i].ld < D || t[i].num_entries <= 0) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding: ld < out_dim or num_entries <= 0");
  }
  (void)who;
  return FFH_OK;
}

bool can_vec4(const ffh_emb_table* t, int nt, int D) {
  if (D % 4) return false;
  for (int i = 0; i < nt; i++)
    if (!aligned16(t[i].weight) || !aligned16(t[i].io) || (t[i].ld % 4)) return false;
  return true;
}

}  // namespace

// row-wise sharded table: global id -> local id, rows held elsewhere -> the zero row behind the local slice
__global__ __launch_bounds__(256) void emb_localize_kernel(const int64_t* __restrict__ idx, int64_t* __restrict__ local, int64_t n,
                                                           int64_t row_begin, int64_t rows_local) {
  ffh_kernel_prio();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = idx[i] - row_begin;
    local[i] = (r >= 0 && r < rows_local) ? r : rows_local;
  }
}

extern "C" {

int ffh_embedding_fwd_multi(ffh_ctx* c, const ffh_emb_table* tables, int nt, int L, int D, int64_t batch, int aggr, ffh_stream s) {
  int rc = validate_tables(c, tables, nt, L, D, batch, aggr, "embedding_fwd");
  if (rc) return rc;
  if (nt == 0 || batch == 0) return FFH_OK;
  const bool v4 = can_vec4(tables, nt, D);
  const int nvec = v4 ? D / 4 : D;
  if ((nvec + 63) / 64 > kMaxChunks * 64) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "embedding_fwd: out_dim too large");
  EmbArgs a;
  memset(&a, 0, sizeof a);
  for (int i = 0; i < nt; i++) {
    a.t[i] = tables[i];
    // tensor-op mode with a registered twin of the destination: the gather writes the bf16 roundings beside the fp32 rows
    a.out16[i] = (v4 && tables[i].ld % 4 == 0) ? ffh_mirror_of(c, tables[i].io, (size_t)((batch - 1) * tables[i].ld + D) * 4) : nullptr;
    // split mode with a registered image of the destination (rows a whole number of 32-element groups apart): the three terms beside the fp32 rows
    int col0 = 0;
    a.out3[i] = (v4 && tables[i].ld % 32 == 0) ? ffh_planes_of(c, tables[i].io, (size_t)((batch - 1) * tables[i].ld + D) * 4, &col0) : nullptr;
    a.out3c[i] = col0;
    if (a.out3[i] && (col0 & 3)) a.out3[i] = nullptr;
  }
  a.batch = batch; a.ntables = nt; a.L = L; a.D = D; a.aggr = aggr;
  // Cache policy (round 6): the output is written once and read much later by a GEMM -- 436 MB per launch at the Terabyte shape, more than the
  // Infinity Cache holds -- and a row of a table too big to stay cached is touched once per launch: both as NONTEMPORAL accesses, so that they do
  // not evict the rows of the 18 small tables (62 MB) that do live in L2 / Infinity Cache.  One box, interleaved: 139.4 -> 133.7 (stores) / 133.8
  // (loads) / 130.9 us (both) = 0.79 -> 0.84 of 8 TB/s by the algorithmic-bytes formula; the step unchanged.  Same bits.
  a.nt = FFH_LAB_INT("FFH_EMB_NT", 3);
  a.nt_rows = (int64_t)FFH_LAB_INT("FFH_EMB_NT_MB", 64) * (1 << 20) / ((int64_t)D * 4);      // tables above 64 MB
  const int lpr = nvec < 64 ? nvec : 64;
  const int rpw = 64 / lpr;
  constexpr int U = 4;
  const int64_t rows_per_block = 4LL * rpw * U;
  // fill 256 CUs x 8 workgroups across all tables, grid-stride the rest
  int64_t gx = (batch + rows_per_block - 1) / rows_per_block;
  static const int cap_env = FFH_LAB_INT("FFH_EMB_FWD_CAP", 1024);   // A/B switch: workgroups over all tables
  const int64_t cap = cap_env / nt > 0 ? cap_env / nt : 1;
  if (gx > cap) gx = cap;
  dim3 grid((unsigned)gx, (unsigned)nt);
  if (v4) hipLaunchKernelGGL((emb_fwd_kernel<4, U>), grid, dim3(256), 0, as_stream(s), a);
  else hipLaunchKernelGGL((emb_fwd_kernel<1, U>), grid, dim3(256), 0, as_stream(s), a);
  FFH_LAUNCH_CHECK(c, "emb_fwd_kernel");
  return FFH_OK;
}

int ffh_embedding_fwd(ffh_ctx* c, const int64_t* idx, float* out, const float* weight, int L, int D, int64_t batch,
                      int64_t num_entries, int64_t out_ld, int aggr, ffh_stream s) {
  ffh_emb_table t{idx, const_cast<float*>(weight), out, num_entries, out_ld};
  return ffh_embedding_fwd_multi(c, &t, 1, L, D, batch, aggr, s);
}

int ffh_embedding_bwd_dense(ffh_ctx* c, const int64_t* idx, const float* g, float* wg, int L, int D, int64_t batch,
                            int64_t num_entries, int64_t gld, int aggr, ffh_stream s) {
  ffh_emb_table t{idx, wg, const_cast<float*>(g), num_entries, gld};
  int rc = validate_tables(c, &t, 1, L, D, batch, aggr, "embedding_bwd_dense");
  if (rc) return rc;
  if (batch == 0) return FFH_OK;
  hipLaunchKernelGGL(emb_bwd_dense_kernel, dim3(ffh_grid(batch * D, 256, 4096)), dim3(256), 0, as_stream(s),
                     idx, g, wg, L, D, batch, gld, aggr == FFH_AGGR_MODE_AVG ? 1 : 0);
  FFH_LAUNCH_CHECK(c, "emb_bwd_dense_kernel");
  return FFH_OK;
}

int ffh_embedding_localize_rows(ffh_ctx* c, const int64_t* idx, int64_t* local, int64_t n, int64_t row_begin, int64_t rows_local, ffh_stream s) {
  FFH_REQUIRE(c, n >= 0 && row_begin >= 0 && rows_local >= 0 && ((idx && local) || n == 0), "embedding_localize_rows: bad args");
  if (n == 0) return FFH_OK;
  hipLaunchKernelGGL(emb_localize_kernel, dim3(ffh_grid(n, 256)), dim3(256), 0, as_stream(s), idx, local, n, row_begin, rows_local);
  FFH_LAUNCH_CHECK(c, "embedding_localize_rows");
  return FFH_OK;
}

size_t ffh_embedding_bwd_workspace_bytes(int nt, int L, int D, int64_t batch) {
  if (nt <= 0 || L <= 0 || D <= 0 || batch <= 0) return 0;
  return bwd_layout(nt, L, D, batch).total;
}

// The fused update in two phases: the stable sort of (row id, position) needs the indices only, so a caller that knows them
// early (the DLRM step: at the gather) can run it off the critical path (ffh_embedding_bwd_sort_multi) and do the part that
// needs the output gradients -- segmented reduce, folds, the SGD step -- when they exist (ffh_embedding_bwd_sgd_apply_multi).
// Same launches in the same order on the same workspace: the fused entry is both phases back to back.
// `opt` / `states`: the row rule (ffh_sparse_opt) and the per-table optimizer state it updates; null opt = plain SGD with `lr`
static int emb_bwd_phases(ffh_ctx* c, const ffh_emb_table* tables, int nt, int L, int D, int64_t batch,
                          int aggr, float lr, ffh_stream s, const bool do_sort, const bool do_apply,
                          const ffh_sparse_opt* opt = nullptr, const ffh_emb_state* states = nullptr) {
  int rc = validate_tables(c, tables, nt, L, D, batch, aggr, "embedding_bwd_sgd_fused");
  if (rc) return rc;
  OptP op{};
  op.lr = lr;
  // (round 6) the rows of a table above 64 MB are read and written back NONTEMPORAL by the plain-SGD apply step: a row of such a table is touched once
  // per launch and must not evict the small tables' rows from L2 / Infinity Cache (as in the gather): 26 tables x 32768 lookups 198.0 -> 188.2 us
  // = 0.83 -> 0.875 of 8 TB/s, interleaved on one box; same bits
  op.nt_rows = (int64_t)FFH_LAB_INT("FFH_EMB_APPLY_NT_MB", 64) * (int64_t)(1 << 20) / ((int64_t)D * 4);
  int kind = FFH_SPARSE_OPT_SGD;
  if (opt && do_apply) {
    kind = opt->kind;
    if (kind != FFH_SPARSE_OPT_SGD && kind != FFH_SPARSE_OPT_SGD_MOMENTUM && kind != FFH_SPARSE_OPT_ADAM)
      return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: unknown ffh_sparse_opt.kind");
    op.lr = opt->lr; op.wd = opt->weight_decay; op.mom = opt->momentum; op.nesterov = opt->nesterov ? 1 : 0;
    op.b1 = opt->beta1; op.b2 = opt->beta2; op.eps = opt->epsilon; op.omb1 = 1.0f - opt->beta1; op.omb2 = 1.0f - opt->beta2;
    if (kind == FFH_SPARSE_OPT_SGD && (op.wd != 0.0f || op.mom != 0.0f)) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: FFH_SPARSE_OPT_SGD takes no weight decay / momentum (use FFH_SPARSE_OPT_SGD_MOMENTUM)");
    const bool need0 = kind == FFH_SPARSE_OPT_ADAM || (kind == FFH_SPARSE_OPT_SGD_MOMENTUM && op.mom > 0.0f);
    for (int i = 0; i < nt && batch > 0; i++) {
      if (need0 && (!states || !states[i].s0)) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: optimizer state (s0) missing");
      if (kind == FFH_SPARSE_OPT_ADAM && !states[i].s1) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt: optimizer state (s1) missing");
    }
  }
  if (nt == 0 || batch == 0) return FFH_OK;
  const int64_t N = batch * L;
  if (N >= (1LL << 31)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "embedding_bwd_sgd_fused: batch*in_dim >= 2^31");
  int64_t maxR = 1;
  for (int i = 0; i < nt; i++) maxR = tables[i].num_entries > maxR ? tables[i].num_entries : maxR;
  if (maxR > (1LL << 32)) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "embedding_bwd_sgd_fused: num_entries > 2^32");
  bool v4 = can_vec4(tables, nt, D);
  for (int i = 0; i < nt && v4 && kind != FFH_SPARSE_OPT_SGD; i++)
    v4 = (!states[i].s0 || aligned16(states[i].s0)) && (!states[i].s1 || aligned16(states[i].s1));
  const int nvec = v4 ? D / 4 : D;
  if ((nvec + 63) / 64 > kMaxChunks * 64) return ffh_fail(c, FFH_ERR_UNSUPPORTED, "embedding_bwd_sgd_fused: out_dim too large");
  const BwdLayout lay = bwd_layout(nt, L, D, batch);
  if (!c->ws || c->ws_bytes < lay.total) return ffh_fail(c, FFH_ERR_WORKSPACE, "embedding_bwd_sgd_fused: workspace too small (ffh_embedding_bwd_workspace_bytes)");
  char* ws = (char*)c->ws;
  if (!aligned16(ws)) return ffh_fail(c, FFH_ERR_WORKSPACE, "embedding_bwd_sgd_fused: workspace must be 16-byte aligned");
  // The apply phase CONSUMES what the sort phase left in the workspace: besides the sorted list, the arrival counters and level-1
  // slots that only the sort's first histogram pass clears -- a second apply on the same sort would find them used (no workgroup
  // would be "last", rows whose runs cross tiles would silently miss their update).  The two-call form is therefore one-shot: the
  // sort notes (workspace, shape) in the ctx, the apply requires and clears that note; the fused call needs no note but spoils one
  // that sits on the workspace it overwrites.
  const int64_t sig[4] = {nt, L, D, batch};
  if (do_sort && do_apply) {
    if (c->emb_sorted_ws == c->ws) c->emb_sorted_ws = nullptr;
  } else if (do_sort) {
    c->emb_sorted_ws = c->ws;
    memcpy(c->emb_sorted_sig, sig, sizeof sig);
  } else {
    if (c->emb_sorted_ws != c->ws || memcmp(c->emb_sorted_sig, sig, sizeof sig) != 0)
      return ffh_fail(c, FFH_ERR_WORKSPACE, "embedding_bwd_sgd_apply_multi: no fresh ffh_embedding_bwd_sort_multi of the same tables / batch on this ctx's "
                                           "workspace (the apply phase consumes the sort: one apply per sort)");
    c->emb_sorted_ws = nullptr;
  }

  if (N <= kSmallMax) {
    // small-batch path: one launch, one workgroup per table (see emb_sgd_small_kernel): the sort lives inside it
    snprintf(c->emb_route, sizeof c->emb_route, "small");
    if (!do_apply) return FFH_OK;
    int bits_s = 1;
    while (bits_s < 32 && ((maxR - 1) >> bits_s) != 0) bits_s++;
    const int passes_s = (bits_s + kMaxRadixBits - 1) / kMaxRadixBits;
    const int rb_s = (bits_s + passes_s - 1) / passes_s;
    SmallArgs sm;
    memset(&sm, 0, sizeof sm);
    for (int i = 0; i < nt; i++) {
      sm.t[i] = tables[i];
      int tb = 1;
      while (tb < 32 && ((tables[i].num_entries - 1) >> tb) != 0) tb++;
      sm.npass[i] = (uint8_t)((tb + rb_s - 1) / rb_s);
    }
    sm.kp = (uint2*)(ws + lay.kp_a);
    sm.partial0 = (float*)(ws + lay.partial); sm.meta0 = (uint2*)(ws + lay.meta);
    sm.partial1 = (float*)(ws + lay.partial1); sm.meta1 = (uint2*)(ws + lay.meta1);
    sm.N = N; sm.nch0 = lay.nchunks; sm.nch1 = lay.nchunks1; sm.rb = rb_s; sm.L = L; sm.D = D;
    sm.avg = aggr == FFH_AGGR_MODE_AVG ? 1 : 0; sm.op = op;
    for (int i = 0; i < nt && kind != FFH_SPARSE_OPT_SGD; i++) { sm.s0[i] = states[i].s0; sm.s1[i] = states[i].s1; }
    int tile = (int)((N + kSmallRedParts - 1) / kSmallRedParts);          // one tile per 256-thread team
    tile = (tile + FFH_EMB_CHUNK - 1) / FFH_EMB_CHUNK * FFH_EMB_CHUNK;
    sm.tile = tile < FFH_EMB_CHUNK ? FFH_EMB_CHUNK : tile;
#define FFH_SMALL(OPTV)                                                                                               \
    { if (v4) hipLaunchKernelGGL((emb_sgd_small_kernel<4, OPTV>), dim3(nt), dim3(kSmallThreads), 0, as_stream(s), sm);  \
      else hipLaunchKernelGGL((emb_sgd_small_kernel<1, OPTV>), dim3(nt), dim3(kSmallThreads), 0, as_stream(s), sm); }
    if (kind == FFH_SPARSE_OPT_SGD) FFH_SMALL(0) else if (kind == FFH_SPARSE_OPT_SGD_MOMENTUM) FFH_SMALL(1) else FFH_SMALL(2)
#undef FFH_SMALL
    FFH_LAUNCH_CHECK(c, "emb_sgd_small_kernel");
    return FFH_OK;
  }
  // radix plan: digits of <= 9 bits covering bit_length(maxR-1); a table only runs the passes its own ids need
  int bits = 1;
  while (bits < 32 && ((maxR - 1) >> bits) != 0) bits++;
  // bucket form (msd_window): one pass on the top digit, the rest inside the apply launch.  Where the LSD form would need >= 2 passes
  // and a bucket averages <= 128 entries (N <= 64 K at 512 buckets); the digit: ~32 entries per bucket, 4 .. 9 bits
  static const int msd_env = FFH_LAB_INT("FFH_EMB_MSD", 1);          // A/B switch: 0 = never, 1 = by shape, 2 = wherever it is valid
  static const int msd_max_tables = FFH_LAB_INT("FFH_EMB_MSD_MAX_TABLES", FFH_MAX_TABLES);      // (first rule: <= 8 tables; at 26 tables x 32768 lookups the form takes 196 instead of 238 us)
  int mb = 4;
  while (mb < kMaxRadixBits && (N >> (mb + 1)) >= 32) mb++;
  const bool msd = msd_env != 0 && bits > kMaxRadixBits && N <= 65536 && (N >> mb) <= 128 && (msd_env == 2 || nt <= msd_max_tables) &&
                   (size_t)lay.nblk * kMaxRadix >= (size_t)lay.nchunks1;
  const int passes = msd ? 1 : (bits + kMaxRadixBits - 1) / kMaxRadixBits;
  const int rb = msd ? mb : (bits + passes - 1) / passes;
  if (msd) snprintf(c->emb_route, sizeof c->emb_route, "buckets:bits=%d", rb); else snprintf(c->emb_route, sizeof c->emb_route, "lsd:passes=%d", passes);

  SortArgs sa;
  memset(&sa, 0, sizeof sa);
  RedArgs ra;
  memset(&ra, 0, sizeof ra);
  for (int i = 0; i < nt; i++) {
    sa.idx[i] = tables[i].idx;
    int tb = 1;
    while (tb < 32 && ((tables[i].num_entries - 1) >> tb) != 0) tb++;
    const int np = msd ? 1 : (tb + rb - 1) / rb;
    sa.npass[i] = (uint8_t)np;
    ra.parity[i] = (uint8_t)(np & 1);
    sa.shift_t[i] = ra.shift_t[i] = (uint8_t)(msd && tb > rb ? tb - rb : 0);
  }
  sa.msd = msd ? 1 : 0;
  sa.bstart = (uint32_t*)(ws + lay.bstart);
  ra.bstart = sa.bstart; ra.nextkey = (uint32_t*)(ws + lay.hist); ra.radix = 1 << rb;
  sa.hist = (uint32_t*)(ws + lay.hist);
  sa.N = N; sa.nblk = lay.nblk; sa.bits = rb;
  sa.clear[0] = (uint32_t*)(ws + lay.meta1); sa.nclear[0] = 4 * lay.nchunks1;     // uint2 slots, two per 1024-block
  sa.clear[1] = (uint32_t*)(ws + lay.arrive); sa.nclear[1] = lay.nchunks1 + 1;
  uint2* kbuf[2] = {(uint2*)(ws + lay.kp_a), (uint2*)(ws + lay.kp_b)};
  dim3 sgrid((unsigned)lay.nblk, (unsigned)nt);
  const int E = sort_per_thread(nt, N);
  for (int p = 0; p < passes && do_sort; p++) {
    sa.shift = p * rb;
    sa.pass = p;
    // pass p reads buffer p%2 (pass 0: the int64 ids) and writes buffer (p+1)%2
    sa.src = kbuf[p & 1];
    sa.dst = kbuf[(p + 1) & 1];
#define FFH_SORT_PASS(FIRSTV, EV)                                                                              \
    hipLaunchKernelGGL((radix_hist_kernel<FIRSTV, EV>), sgrid, dim3(kSortThreads), 0, as_stream(s), sa);      \
    hipLaunchKernelGGL((radix_scatter_kernel<FIRSTV, EV>), sgrid, dim3(kSortThreads), 0, as_stream(s), sa);
    if (p == 0) {
      switch (E) { case 1: FFH_SORT_PASS(true, 1) break; case 2: FFH_SORT_PASS(true, 2) break; case 4: FFH_SORT_PASS(true, 4) break; default: FFH_SORT_PASS(true, 8) break; }
    } else {
      switch (E) { case 1: FFH_SORT_PASS(false, 1) break; case 2: FFH_SORT_PASS(false, 2) break; case 4: FFH_SORT_PASS(false, 4) break; default: FFH_SORT_PASS(false, 8) break; }
    }
#undef FFH_SORT_PASS
  }
  FFH_LAUNCH_CHECK(c, "radix sort");
  if (!do_apply) return FFH_OK;

  for (int i = 0; i < nt; i++) ra.t[i] = tables[i];
  ra.kp[0] = kbuf[0]; ra.kp[1] = kbuf[1];
  ra.tile = reduce_tile(nt, N);
  {
    static const int msd_tile = FFH_LAB_INT("FFH_EMB_MSD_TILE", 0);      // A/B switch: the bucket form's tile
    if (msd && msd_tile >= 128 && msd_tile <= kRedTile && (msd_tile & (msd_tile - 1)) == 0) ra.tile = msd_tile;
  }
  ra.partial = (float*)(ws + lay.partial);
  ra.meta = (uint2*)(ws + lay.meta);
  ra.N = N; ra.nchunks = lay.nchunks; ra.L = L; ra.D = D;
  ra.avg = aggr == FFH_AGGR_MODE_AVG ? 1 : 0;
  ra.op = op;
  for (int i = 0; i < nt && kind != FFH_SPARSE_OPT_SGD; i++) { ra.s0[i] = states[i].s0; ra.s1[i] = states[i].s1; }
  ra.partial1 = (float*)(ws + lay.partial1);
  ra.meta1 = (uint2*)(ws + lay.meta1); ra.nchunks1 = lay.nchunks1;
  ra.arrive = (uint32_t*)(ws + lay.arrive);
  // segmented sums + both folds (32-block partials -> 1024-block partials -> row totals) + the SGD step: one launch
  dim3 rgrid((unsigned)((N + ra.tile - 1) / ra.tile), (unsigned)nt);
#define FFH_RED(OPTV, MSDV)                                                                                                 \
  { if (v4) hipLaunchKernelGGL((emb_sgd_reduce_kernel<4, OPTV, MSDV>), rgrid, dim3(kRedThreads), 0, as_stream(s), ra);        \
    else hipLaunchKernelGGL((emb_sgd_reduce_kernel<1, OPTV, MSDV>), rgrid, dim3(kRedThreads), 0, as_stream(s), ra); }
#ifdef FFH_MSD_TIMING
  static unsigned long long* dbg_buf = nullptr;
  if (!dbg_buf) hipMalloc(&dbg_buf, 8 * 8 * 65536);
  ra.dbg = dbg_buf;
#endif
  if (msd) { if (kind == FFH_SPARSE_OPT_SGD) FFH_RED(0, true) else if (kind == FFH_SPARSE_OPT_SGD_MOMENTUM) FFH_RED(1, true) else FFH_RED(2, true) }
  else { if (kind == FFH_SPARSE_OPT_SGD) FFH_RED(0, false) else if (kind == FFH_SPARSE_OPT_SGD_MOMENTUM) FFH_RED(1, false) else FFH_RED(2, false) }
#undef FFH_RED
  FFH_LAUNCH_CHECK(c, "emb_sgd_reduce/fold");
#ifdef FFH_MSD_TIMING
  {
    static int calls = 0;
    if (++calls % 23 == 0) {
      hipDeviceSynchronize();
      const int nwg = (int)(rgrid.x * rgrid.y);
      std::vector<unsigned long long> h((size_t)nwg * 8);
      hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull, t2max = 0; double s1 = 0, s2 = 0, start = 0;
      for (int i = 0; i < nwg; i++) { t0 = h[i * 8] < t0 ? h[i * 8] : t0; }
      for (int i = 0; i < nwg; i++) { s1 += (double)(h[i * 8 + 1] - h[i * 8]); s2 += (double)(h[i * 8 + 2] - h[i * 8 + 1]); start += (double)(h[i * 8] - t0); t2max = h[i * 8 + 2] > t2max ? h[i * 8 + 2] : t2max; }
      fprintf(stderr, "[msd timing] %d workgroups (msd %d): start after first %.2f us avg, window %.2f us avg, reduce %.2f us avg, last reduce end %.2f us after first start (100 MHz ticks)\n",
              nwg, (int)msd, start / nwg / 100.0, s1 / nwg / 100.0, s2 / nwg / 100.0, (double)(t2max - t0) / 100.0);
    }
  }
#endif
  return FFH_OK;
}

int ffh_embedding_bwd_sgd_fused_multi(ffh_ctx* c, const ffh_emb_table* tables, int nt, int L, int D, int64_t batch,
                                      int aggr, float lr, ffh_stream s) {
  return emb_bwd_phases(c, tables, nt, L, D, batch, aggr, lr, s, true, true);
}

int ffh_embedding_bwd_sort_multi(ffh_ctx* c, const ffh_emb_table* tables, int nt, int L, int D, int64_t batch, ffh_stream s) {
  return emb_bwd_phases(c, tables, nt, L, D, batch, FFH_AGGR_MODE_SUM, 0.0f, s, true, false);
}

int ffh_embedding_bwd_sgd_apply_multi(ffh_ctx* c, const ffh_emb_table* tables, int nt, int L, int D, int64_t batch,
                                      int aggr, float lr, ffh_stream s) {
  return emb_bwd_phases(c, tables, nt, L, D, batch, aggr, lr, s, false, true);
}

int ffh_embedding_bwd_opt_fused_multi(ffh_ctx* c, const ffh_emb_table* tables, const ffh_emb_state* states, int nt, int L, int D, int64_t batch,
                                      int aggr, const ffh_sparse_opt* opt, ffh_stream s) {
  if (!opt) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt_fused_multi: null ffh_sparse_opt");
  return emb_bwd_phases(c, tables, nt, L, D, batch, aggr, opt->lr, s, true, true, opt, states);
}

int ffh_embedding_bwd_opt_apply_multi(ffh_ctx* c, const ffh_emb_table* tables, const ffh_emb_state* states, int nt, int L, int D, int64_t batch,
                                      int aggr, const ffh_sparse_opt* opt, ffh_stream s) {
  if (!opt) return ffh_fail(c, FFH_ERR_BAD_ARG, "embedding_bwd_opt_apply_multi: null ffh_sparse_opt");
  return emb_bwd_phases(c, tables, nt, L, D, batch, aggr, opt->lr, s, false, true, opt, states);
}

int ffh_embedding_bwd_sgd_fused(ffh_ctx* c, const int64_t* idx, const float* g, float* weight, int L, int D, int64_t batch,
                                int64_t num_entries, int64_t gld, int aggr, float lr, ffh_stream s) {
  ffh_emb_table t{idx, weight, const_cast<float*>(g), num_entries, gld};
  return ffh_embedding_bwd_sgd_fused_multi(c, &t, 1, L, D, batch, aggr, lr, s);
}

}  // extern "C"
