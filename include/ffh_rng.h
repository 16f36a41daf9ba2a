/* ffh_rng.h -- the counter-based generator behind ffh_init_uniform / ffh_gen_*.
 *
 * The reference draws weights with cuRAND [ref: src/runtime/initializer_kernel.cu:24-207]
 * and synthetic inputs with unseeded std::rand() [ref: examples/cpp/DLRM/dlrm.cc:413-420];
 * neither stream is reproducible, so parity always runs on injected or
 * counter-generated data.  This header fixes ONE integer
 * hash so that the HIP backend, the CPU oracle and the C++ host shim produce
 * bit-identical tensors from (seed, element index) with no state.
 * Distributions follow the reference: uniform ids in [0, R), dense in [0,1),
 * labels in {0,1}, uniform weights in [lo, hi).
 */
#ifndef FFH_RNG_H_
#define FFH_RNG_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define FFH_HD __host__ __device__ static inline
#else
#define FFH_HD static inline
#endif

/* splitmix64 finaliser */
FFH_HD uint64_t ffh_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* 64 random bits for element `i` of stream `seed` */
FFH_HD uint64_t ffh_hash(uint64_t seed, uint64_t i) {
  return ffh_mix64(ffh_mix64(seed) + i);
}

/* 24-bit uniform in [0,1): exact in fp32 */
FFH_HD float ffh_u24(uint64_t h) {
  return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}

/* lo + (hi-lo)*u with one fused rounding (fmaf on both sides) */
FFH_HD float ffh_uniform(uint64_t h, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fmaf_rn(ffh_u24(h), hi - lo, lo);
#else
  return __builtin_fmaf(ffh_u24(h), hi - lo, lo);
#endif
}

FFH_HD int64_t ffh_index(uint64_t h, int64_t num_entries) {
  return (int64_t)(h % (uint64_t)num_entries);
}

FFH_HD float ffh_bernoulli(uint64_t h) {
  return (float)((h >> 33) & 1ULL);
}

#endif /* FFH_RNG_H_ */
