// probe: semantics of v_dot2c_f32_bf16 on gfx950 (used for the bf16x3 operand split in csrc/gemm.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* o) {
  const int t = threadIdx.x;
  const float a = x[2 * t], b = x[2 * t + 1];
  unsigned hi;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  const bf16x2 hv = __builtin_bit_cast(bf16x2, hi);
  o[6 * t + 0] = __uint_as_float(hi << 16);
  o[6 * t + 1] = __uint_as_float(hi & 0xFFFF0000u);
  o[6 * t + 2] = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, 0x8000BF80u), a, false);
  o[6 * t + 3] = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, 0xBF808000u), b, false);
  o[6 * t + 4] = a - __uint_as_float(hi << 16);
  o[6 * t + 5] = b - __uint_as_float(hi & 0xFFFF0000u);
}
int main() {
  float hx[16] = {1.2345678f, -3.3333333f, 1000.123f, 1e-3f, 0.5f, 0.75f, 3.1415927f, 2.7182818f, 1e-20f, 1e20f, -7.7f, 8.8f, 0.f, 1.f, 123456.7f, -0.001234f};
  float *dx, *dout; hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, 8 * 6 * 4);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  k<<<1, 8>>>(dx, dout);
  float ho[48]; hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; ++t)
    printf("x=(%.9g, %.9g) hi=(%.9g, %.9g) dot2=(%.9g, %.9g) sub=(%.9g, %.9g)\n", hx[2 * t], hx[2 * t + 1], ho[6 * t], ho[6 * t + 1],
           ho[6 * t + 2], ho[6 * t + 3], ho[6 * t + 4], ho[6 * t + 5]);
  return 0;
}
