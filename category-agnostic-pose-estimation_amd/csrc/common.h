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

// ---- wave-wide reductions on the VALU: DPP inside a 16-lane row (quad xor 1, xor 2, half-row mirror, row mirror), then
// v_permlane16_swap / v_permlane32_swap across rows (new on gfx950).  The shuffle form (6 dependent ds_bpermute = LDS crossbar
// round trips) put ~1 us of latency into every LayerNorm row and softmax; this one is ~12 VALU instructions.
// The swaps are inline asm: `__builtin_amdgcn_permlane32_swap` on hipcc 7.2 adds result 0 to itself when both results feed
// one add (tools/lab/permlane_probe2.hip).  All lanes end up with the result, like the shuffle butterflies they replace.
template <int CTRL>
__device__ __forceinline__ float cape_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void cape_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void cape_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float wave_sum(float v) {
  v += cape_dpp<0xB1>(v); v += cape_dpp<0x4E>(v); v += cape_dpp<0x141>(v); v += cape_dpp<0x140>(v);   // 16-lane row sums
  float a = v, b = v;
  cape_swap16(a, b);                                              // a = {r0, r0, r2, r2}, b = {r1, r1, r3, r3}
  v = a + b;
  a = v; b = v;
  cape_swap32(a, b);                                              // a = {lo, lo}, b = {hi, hi}
  return a + b;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, cape_dpp<0xB1>(v)); v = fmaxf(v, cape_dpp<0x4E>(v)); v = fmaxf(v, cape_dpp<0x141>(v)); v = fmaxf(v, cape_dpp<0x140>(v));
  float a = v, b = v;
  cape_swap16(a, b);
  v = fmaxf(a, b);
  a = v; b = v;
  cape_swap32(a, b);
  return fmaxf(a, b);
}
