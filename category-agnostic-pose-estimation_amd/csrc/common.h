// common.h -- shared device/host helpers for libcape_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cape_hip.h"

int cape_set_error(const char* fmt, ...);

#define CAPE_REQUIRE(cond, ...)                      \
  do {                                               \
    if (!(cond)) return cape_set_error(__VA_ARGS__); \
  } while (0)

#define CAPE_LAUNCH_CHECK(name)                                                      \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) return cape_set_error("%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline hipStream_t as_stream(cape_stream_t s) { return (hipStream_t)s; }

// ---- counter-based RNG for dropout: pure function of (seed, step, stream id, element index) ----
__device__ __forceinline__ uint32_t cape_rng_u32(uint64_t seed, uint64_t step, uint32_t stream, uint64_t idx) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (step + 1) + 0xD1342543DE82EF95ull * (uint64_t)(stream + 1);
  x ^= idx * 0xA0761D6478BD642Full;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
// keep-decision: keep with probability 1-p
__device__ __forceinline__ bool cape_keep(uint64_t seed, uint64_t step, uint32_t stream, uint64_t idx, uint32_t thresh) {
  return cape_rng_u32(seed, step, stream, idx) >= thresh;
}
static inline uint32_t cape_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
