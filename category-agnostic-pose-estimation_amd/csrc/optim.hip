// optim.hip -- AdamW over flat parameter arenas with global-norm clipping, all scalars on device
// (graph-replay safe): torch.optim.AdamW semantics (train_cape_episodic.py:527-538) after
// torch.nn.utils.clip_grad_norm_(max_norm) (engine_cape.py:240-246).  Pure HBM streaming:
// 4 reads + 3 writes of 4 bytes per parameter.
#include "common.h"

namespace {

// Sum of squares as per-block PARTIAL sums: exactly SUMSQ_PARTS blocks of 16 waves, block b writes out[b] (blocks beyond the data
// write 0).  No float atomics: a reduction whose order depends on block arrival gave two data-parallel replicas -- same all-reduced
// gradient -- clip coefficients one ulp apart, and their parameters drifted from each other step by step (tools/ddp_rehearsal.py).
// The consumer (adamw_kernel) adds the partials in a fixed order.
constexpr int SUMSQ_PARTS = CAPE_SUMSQ_PARTS;
__global__ void __launch_bounds__(1024) sumsq_kernel(const float* g, long long n, float* out) {
  __shared__ float sh[16];
  float s = 0.f;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += sh[k];
    out[blockIdx.x] = a;
  }
}

__global__ void __launch_bounds__(256) adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr,
                                                     float b1, float b2, float eps, float wd, float max_norm,
                                                     const float* sumsq, int n_parts, const int64_t* step_count, const float* lr_dev) {
  if (lr_dev) lr = lr_dev[0];                                // the schedule's value of this step, kept on the device (replayed graphs)
  float coef = 1.f;
  if (max_norm > 0.f) {
    // global gradient norm from the partial sums, the same fixed order in every block (and on every rank)
    __shared__ float tot;
    if (threadIdx.x < 64) {
      float a = 0.f;
      for (int k = threadIdx.x; k < n_parts; k += 64) a += sumsq[k];
      a = wave_sum(a);
      if (threadIdx.x == 0) tot = a;
    }
    __syncthreads();
    coef = fminf(1.f, max_norm / (sqrtf(tot) + 1e-6f));
  }
  const float t = (float)step_count[0];
  const float bc1 = 1.f - powf(b1, t);
  const float bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  const float decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * decay;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);     // lerp
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    pi -= step_size * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi;
  }
}

__global__ void step_inc_kernel(int64_t* s) { s[0] += 1; }

}  // namespace

extern "C" int cape_sumsq(const float* g, long long n, float* out, cape_stream_t stream) {
  CAPE_REQUIRE(g && out && n >= 0, "cape_sumsq: bad arguments");
  if (n == 0) return 0;
  CAPE_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "cape_sumsq: g must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_PARTS), dim3(1024), 0, as_stream(stream), g, n, out);
  CAPE_LAUNCH_CHECK("cape_sumsq");
  return 0;
}

extern "C" int cape_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float max_norm, const float* sumsq, int n_parts,
                               const int64_t* step_count, const float* lr_dev, cape_stream_t stream) {
  CAPE_REQUIRE(p && g && m && v && step_count && n >= 0, "cape_adamw_step: bad arguments");
  CAPE_REQUIRE(max_norm <= 0.f || (sumsq && n_parts >= 1), "cape_adamw_step: clipping needs the partial sums of cape_sumsq");
  if (n == 0) return 0;
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)b), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, max_norm, sumsq, n_parts, step_count, lr_dev);
  CAPE_LAUNCH_CHECK("cape_adamw_step");
  return 0;
}

extern "C" int cape_step_increment(int64_t* step_count, cape_stream_t stream) {
  CAPE_REQUIRE(step_count, "cape_step_increment: null pointer");
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_count);
  CAPE_LAUNCH_CHECK("cape_step_increment");
  return 0;
}
