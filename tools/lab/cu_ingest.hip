// cu_ingest.hip -- how fast can ONE compute unit pull a 32 MB weight stream (L2 / Infinity Cache resident) into registers?
// Sizing input for csrc/decode_fused.hip (every block streams all decoder weights per step).
//   mode 0: per wave 32 x 16-byte loads issued together, consumed, next batch (the decode kernel's primitive, no overlap)
//   mode 1: two half-batches of 16 loads, one always in flight while the other is consumed
//   mode 2: mode 0 with the next batch issued before the previous is consumed by a dependent shuffle chain (as the kernel does)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/cu_ingest tools/lab/cu_ingest.hip ; run: cu_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(512) ingest(const float4* __restrict__ w, long long n4_per_block, float* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a "row" = 64 lanes x 16 B = 1 KB; wave takes rows wave, wave+8, ...
  const long long rows = n4_per_block / 64;
  float acc = 0.f;
  if (MODE == 0 || MODE == 2) {
    float4 r[32];
    for (long long base = 0; base + 256 <= rows; base += 256) {
#pragma unroll
      for (int i = 0; i < 32; ++i) r[i] = w[(base + wave + 8 * i) * 64 + lane];
      if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += r[i].x + r[i].y + r[i].z + r[i].w;
      if (MODE == 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      }
    }
  } else {
    float4 a[16], b[16];
    long long base = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = w[(base + wave + 8 * i) * 64 + lane];
    for (; base + 256 <= rows; base += 256) {
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = w[(base + 128 + wave + 8 * i) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y + a[i].z + a[i].w;
      __builtin_amdgcn_sched_barrier(0);
      const long long nb = base + 256 + 256 <= rows ? base + 256 : 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = w[(nb + wave + 8 * i) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += b[i].x + b[i].y + b[i].z + b[i].w;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i].x;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main() {
  const long long bytes = 32ll << 20, n4 = bytes / 16;
  float4* w; float* out;
  hipMalloc(&w, bytes); hipMalloc(&out, 4096);
  hipMemset(w, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1, 2, 8, 32, 128, 256};
  for (int mode = 0; mode < 3; ++mode)
    for (int g : grids) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(ingest<0>, dim3(g), dim3(512), 0, 0, w, n4, out);
        if (mode == 1) hipLaunchKernelGGL(ingest<1>, dim3(g), dim3(512), 0, 0, w, n4, out);
        if (mode == 2) hipLaunchKernelGGL(ingest<2>, dim3(g), dim3(512), 0, 0, w, n4, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("mode %d blocks %3d: %8.1f us for 32 MB per block = %6.1f GB/s per CU, %7.1f GB/s aggregate\n", mode, g, best * 1e3, bytes / best / 1e6,
             bytes * (double)g / best / 1e6);
    }
  return 0;
}
