// permlane_probe.hip -- what v_permlane32_swap / v_permlane16_swap / the DPP mirrors actually move, lane by lane (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__global__ void probe(int* out) {
  const int l = threadIdx.x;
  const unsigned a = l, b = 100 + l;
  u2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r.x; out[64 + l] = r.y;
  r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = r.x; out[192 + l] = r.y;
  out[256 + l] = dppi<0x140>(l); out[320 + l] = dppi<0x141>(l); out[384 + l] = dppi<0xB1>(l); out[448 + l] = dppi<0x4E>(l);
}
int main() {
  int* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"swap32.x", "swap32.y", "swap16.x", "swap16.y", "dpp 0x140", "dpp 0x141", "dpp 0xB1", "dpp 0x4E"};
  for (int k = 0; k < 8; ++k) { printf("%-10s:", names[k]); for (int l = 0; l < 64; ++l) printf(" %d", h[64 * k + l]); printf("\n"); }
  return 0;
}
