// checks the reduction helpers of csrc/decode_fused.hip against brute force
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swap32_add(float a, float b) {
  // inline asm: with the builtin, hipcc 7.2 adds result 0 to itself (r.x + r.x) when both results feed one add
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float sum8f(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }
__device__ __forceinline__ float wsum(float v) {
  v = sum8f(v);
  v += dppf<0x140>(v);
  v = swap16_add(v, v);
  return swap32_add(v, v);
}
__device__ __forceinline__ float reduce8(float (&s)[8]) {
  const int lane = threadIdx.x & 63;
  const float t0 = swap32_add(s[0], s[1]), t1 = swap32_add(s[2], s[3]), t2 = swap32_add(s[4], s[5]), t3 = swap32_add(s[6], s[7]);
  const float u0 = swap16_add(t0, t1), u1 = swap16_add(t2, t3);
  const bool up = lane & 8;
  const float keep = up ? u1 : u0, send = up ? u0 : u1;
  return sum8f(keep + dppf<0x140>(send));
}
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  out[l] = wsum((float)l);
  float s[8];
  for (int i = 0; i < 8; ++i) s[i] = (float)(1000 * i + l);
  out[64 + l] = reduce8(s);
}
int main() {
  float* d; (void)hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wsum (expect 2016):"); for (int l = 0; l < 64; ++l) printf(" %.0f", h[l]); printf("\n");
  printf("reduce8 (expect 64000*i8 + 2016, i8 = b5 + 2 b4 + 4 b3):"); for (int l = 0; l < 64; ++l) printf(" %.0f", h[64 + l]); printf("\n");
  return 0;
}
