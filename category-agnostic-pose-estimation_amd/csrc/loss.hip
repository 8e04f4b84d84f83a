// loss.hip -- CAPE criterion for all decoder layers in one launch: class-weighted CE over
// (label != -1) & visible tokens, L1 over (label == coord) & visible tokens, plus the gradients of the
// weighted total (models/cape_losses.py:71-163, roomformer_v2.py:915-953).  One 1024-thread block:
// R*NL <= a few 10^4 rows, so a single CU finishes in microseconds and no inter-block reduction exists.
#include "common.h"

namespace {

__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

__global__ void __launch_bounds__(1024) loss_kernel(const float* logits, const float* coords, const int64_t* labels,
                                                     const uint8_t* vis, const float* target, const float* class_w,
                                                     float w_ce, float w_l1, float loss_scale, float* losses, float* total,
                                                     float* d_logits, float* d_coords, int NL, long long R) {
  __shared__ float sh[16];
  const float cw0 = class_w[0], cw1 = class_w[1], cw2 = class_w[2];
  float ws = 0.f, cn = 0.f;
  for (long long r = threadIdx.x; r < R; r += blockDim.x) {
    const int64_t lb = labels[r];
    if (vis[r] && lb != -1) ws += lb == 0 ? cw0 : (lb == 1 ? cw1 : cw2);
    if (vis[r] && lb == 0) cn += 1.f;
  }
  const float wsum = block_sum(ws, sh);
  const float cnt = block_sum(cn, sh) * 2.f;          // l1 mean runs over both coordinates
  float tot = 0.f;
  for (int l = 0; l < NL; ++l) {
    float ce = 0.f, l1 = 0.f;
    for (long long r = threadIdx.x; r < R; r += blockDim.x) {
      const long long o = (long long)l * R + r;
      const int64_t lb = labels[r];
      const bool v = vis[r] != 0;
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      if (v && lb != -1) {
        const float a = logits[o * 3 + 0], b = logits[o * 3 + 1], c = logits[o * 3 + 2];
        const float m = fmaxf(a, fmaxf(b, c));
        const float ea = expf(a - m), eb = expf(b - m), ec = expf(c - m);
        const float se = ea + eb + ec;
        const float lse = m + logf(se);
        const float w = lb == 0 ? cw0 : (lb == 1 ? cw1 : cw2);
        const float xl = lb == 0 ? a : (lb == 1 ? b : c);
        ce += w * (lse - xl);
        const float k = w / wsum * w_ce * loss_scale;
        g0 = k * (ea / se - (lb == 0 ? 1.f : 0.f));
        g1 = k * (eb / se - (lb == 1 ? 1.f : 0.f));
        g2 = k * (ec / se - (lb == 2 ? 1.f : 0.f));
      }
      d_logits[o * 3 + 0] = g0; d_logits[o * 3 + 1] = g1; d_logits[o * 3 + 2] = g2;
      float h0 = 0.f, h1 = 0.f;
      if (v && lb == 0) {
        const float dx = coords[o * 2 + 0] - target[r * 2 + 0], dy = coords[o * 2 + 1] - target[r * 2 + 1];
        l1 += fabsf(dx) + fabsf(dy);
        const float k = w_l1 * loss_scale / cnt;
        h0 = dx > 0.f ? k : (dx < 0.f ? -k : 0.f);
        h1 = dy > 0.f ? k : (dy < 0.f ? -k : 0.f);
      }
      d_coords[o * 2 + 0] = h0; d_coords[o * 2 + 1] = h1;
    }
    const float ce_t = block_sum(ce, sh) / wsum;
    const float l1_t = block_sum(l1, sh) / cnt;
    if (threadIdx.x == 0) { losses[2 * l] = ce_t; losses[2 * l + 1] = l1_t; }
    tot += w_ce * ce_t + w_l1 * l1_t;
  }
  if (threadIdx.x == 0) total[0] = tot;
}

}  // namespace

extern "C" int cape_loss_fwd_bwd(const float* logits, const float* coords, const int64_t* labels, const uint8_t* vis,
                                 const float* target, const float* class_w, float w_ce, float w_l1, float loss_scale,
                                 float* losses, float* total, float* d_logits, float* d_coords, int NL, long long R,
                                 cape_stream_t stream) {
  CAPE_REQUIRE(logits && coords && labels && vis && target && class_w && losses && total && d_logits && d_coords,
               "cape_loss_fwd_bwd: null pointer");
  CAPE_REQUIRE(NL >= 1 && R >= 1, "cape_loss_fwd_bwd: empty problem");
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, coords, labels, vis, target, class_w,
                     w_ce, w_l1, loss_scale, losses, total, d_logits, d_coords, NL, R);
  CAPE_LAUNCH_CHECK("cape_loss_fwd_bwd");
  return 0;
}
