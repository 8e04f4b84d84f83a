// Micro-benchmark: LDS atomic-add throughput on gfx950 by operand type (conflict-free addresses, 16 waves per CU).
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T, int MODE>
__global__ void __launch_bounds__(1024) k(T* out, int iters, long long* cyc) {
  __shared__ T buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) buf[i] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long long t0 = clock64();
  int idx = (wv * 64 + lane) & 8191;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (idx + u * 1024 + i * 64) & 8191;
      if (MODE == 0) atomicAdd(&buf[a], T(1));
      else buf[a] += T(1);
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = buf[threadIdx.x];
}

template <typename T, int MODE>
void run(const char* name) {
  T* out; long long* cyc; hipMalloc(&out, sizeof(T) * 1024 * 256); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<T, MODE><<<256, 1024>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<T, MODE><<<256, 1024>>>(out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double winstr = 16.0 * iters * 8;      // wave-instructions per CU
  printf("%-28s %8.3f ms  %8.1f clk64/wave-instr  (%.1f ns)\n", name, ms, (double)c / winstr, ms * 1e6 / winstr);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<unsigned, 0>("ds_add_u32 atomic");
  run<float, 0>("ds_add_f32 atomic");
  run<unsigned long long, 0>("ds_add_u64 atomic");
  run<double, 0>("ds_add_f64 atomic");
  run<float, 1>("f32 read+add+write (racy)");
  return 0;
}
