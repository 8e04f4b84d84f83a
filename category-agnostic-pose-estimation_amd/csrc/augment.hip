// augment.hip -- GPU side of the MP-100 loader: raw uint8 crops -> augmented, resized, normalised
// (N, 3, S, S) fp32 batch, two launches per batch whatever its size.
//
// The reference pipes every crop through albumentations on the host cores (datasets/mp100_cape.py:896-950: Affine p 0.7,
// HorizontalFlip p 0.5, ColorJitter(0.3, 0.3, 0.3, 0.1) p 0.6, OneOf(GaussNoise, GaussianBlur 3-7, MotionBlur 5) p 0.3,
// Resize 512).  Here the DataLoader workers only decode, crop and draw a *plan* (datasets/transforms.py: the few random numbers
// of those transforms); the pixels are made on the GPU from the raw crop, so 8 ranks do not queue on the host's cores:
//   cape_augment_warp    per image: aug[y][x] = bilinear sample of the uint8 crop through the plan's 2x3 map (affine about
//                        the centre + flip, zero padding: cv2 BORDER_CONSTANT), fp32 HWC at crop resolution, and the sum of its
//                        grey values (the contrast jitter pivots on the image's mean grey);
//   cape_augment_finish  per output pixel: bilinear resize (cv2 INTER_LINEAR: half-pixel centres, edge clamp) of the
//                        colour-jittered image -- brightness / contrast / saturation / hue in the plan's order, clipped to
//                        [0, 1] after each like the uint8 pipeline -- optionally convolved with the plan's k x k kernel
//                        (Gaussian or motion blur, reflect-101 border) or perturbed by Gaussian noise, then (x - mean) / std.
// Both are pure gathers: a thread owns an output element, crops of one batch are walked by blockIdx.y.  The same arithmetic
// runs on the host in datasets/transforms.apply_plan_host (torch CPU); tests compare the two pixel for pixel.
#include "common.h"

namespace {

struct AugItem {                         // one image of the batch (device array; layout mirrored by hip/lib.AugItem)
  const uint8_t* src;                    // (h, w, 3) uint8
  float* aug;                            // (h, w, 3) fp32 workspace
  float* out;                            // (3, S, S) fp32
  float* stat;                           // [1]: sum of grey values of `aug` (zeroed by the caller)
  int h, w;
  float M[6];                            // aug pixel index (x, y) -> source pixel index: sx = M0 x + M1 y + M2, sy = M3 x + M4 y + M5
  int color_on, order[4];                // order[i] in {0 brightness, 1 contrast, 2 saturation, 3 hue}
  float bright, contrast, sat, hue;
  int mode;                              // 0 none, 1 Gaussian noise, 2 k x k convolution
  float noise_std; uint32_t seed;
  int blur_k; float blur_w[49];
};

__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float grey(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

__global__ void __launch_bounds__(256) augment_warp_kernel(const AugItem* items) {
  __shared__ float part[4];
  const AugItem& it = items[blockIdx.y];
  const int h = it.h, w = it.w;
  const int npix = h * w;
  float gsum = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    const int y = p / w, x = p - y * w;
    const float sx = it.M[0] * x + it.M[1] * y + it.M[2];
    const float sy = it.M[3] * x + it.M[4] * y + it.M[5];
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float ax = sx - fx0, ay = sy - fy0;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xx = x0 + dx, yy = y0 + dy;
        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;                      // BORDER_CONSTANT, fill 0
        const float wt = (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay);
        const uint8_t* s = it.src + ((long long)yy * w + xx) * 3;
        acc[0] += wt * s[0]; acc[1] += wt * s[1]; acc[2] += wt * s[2];
      }
    const float r = acc[0] * (1.f / 255.f), g = acc[1] * (1.f / 255.f), b = acc[2] * (1.f / 255.f);
    float* o = it.aug + (long long)p * 3;
    o[0] = r; o[1] = g; o[2] = b;
    gsum += grey(r, g, b);
  }
  gsum = wave_sum(gsum);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = gsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(it.stat, part[0] + part[1] + part[2] + part[3]);
}

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
  // RGB -> HSV, h += shift (fraction of the circle), -> RGB
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  const float d = mx - mn;
  float hh = 0.f;
  if (d > 0.f) {
    if (mx == r) hh = (g - b) / d + (g < b ? 6.f : 0.f);
    else if (mx == g) hh = (b - r) / d + 2.f;
    else hh = (r - g) / d + 4.f;
    hh *= (1.f / 6.f);
  }
  const float s = mx > 0.f ? d / mx : 0.f, v = mx;
  hh = hh + shift;
  hh -= floorf(hh);
  const float h6 = hh * 6.f;
  const int i = (int)floorf(h6) % 6;
  const float f = h6 - floorf(h6);
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// colour chain of one pixel; `mean_grey` = mean grey of the warped image (scaled by the brightness factor when brightness
// comes before contrast in the order -- clipping and the hue step are ignored in that pivot, on the host too)
__device__ __forceinline__ void jitter(const AugItem& it, float mean_grey, float& r, float& g, float& b) {
  if (!it.color_on) return;
  bool bright_done = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int op = it.order[k];
    if (op == 0) {
      r = clip01(r * it.bright); g = clip01(g * it.bright); b = clip01(b * it.bright);
      bright_done = true;
    } else if (op == 1) {
      const float m = mean_grey * (bright_done ? it.bright : 1.f);
      r = clip01((r - m) * it.contrast + m); g = clip01((g - m) * it.contrast + m); b = clip01((b - m) * it.contrast + m);
    } else if (op == 2) {
      const float gy = grey(r, g, b);
      r = clip01((r - gy) * it.sat + gy); g = clip01((g - gy) * it.sat + gy); b = clip01((b - gy) * it.sat + gy);
    } else {
      hue_shift(r, g, b, it.hue);
    }
  }
}

__device__ __forceinline__ int reflect101(int i, int n) {           // cv2 BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

__device__ __forceinline__ float gauss_noise(uint32_t seed, uint32_t ch, uint64_t idx) {
  const uint32_t a = cape_rng_u32(seed, 0, ch, 2 * idx), c = cape_rng_u32(seed, 0, ch, 2 * idx + 1);
  const float u1 = ((float)a + 1.f) * (1.f / 4294967296.f), u2 = (float)c * (1.f / 4294967296.f);
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// colour-jittered (and blurred / noised) value of aug pixel (x, y)
__device__ __forceinline__ void aug_pixel(const AugItem& it, float mean_grey, int x, int y, float (&v)[3]) {
  const int h = it.h, w = it.w;
  if (it.mode == 2) {
    const int k = it.blur_k, c = k >> 1;
    v[0] = v[1] = v[2] = 0.f;
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < k; ++i) {
        const float wt = it.blur_w[j * k + i];
        if (wt == 0.f) continue;
        const int xx = reflect101(x + i - c, w), yy = reflect101(y + j - c, h);
        const float* s = it.aug + ((long long)yy * w + xx) * 3;
        float r = s[0], g = s[1], b = s[2];
        jitter(it, mean_grey, r, g, b);
        v[0] += wt * r; v[1] += wt * g; v[2] += wt * b;
      }
    return;
  }
  const float* s = it.aug + ((long long)y * w + x) * 3;
  float r = s[0], g = s[1], b = s[2];
  jitter(it, mean_grey, r, g, b);
  if (it.mode == 1) {
    const uint64_t idx = (uint64_t)y * w + x;
    r = clip01(r + it.noise_std * gauss_noise(it.seed, 0, idx));
    g = clip01(g + it.noise_std * gauss_noise(it.seed, 1, idx));
    b = clip01(b + it.noise_std * gauss_noise(it.seed, 2, idx));
  }
  v[0] = r; v[1] = g; v[2] = b;
}

__global__ void __launch_bounds__(256) augment_finish_kernel(const AugItem* items, int S, const float* mean, const float* stdv) {
  const AugItem& it = items[blockIdx.y];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int v_ = p / S, u = p - v_ * S;
  const int h = it.h, w = it.w;
  const float mean_grey = it.stat[0] / (float)(h * w);
  // cv2 INTER_LINEAR: source coordinate of the output pixel centre, clamped at the borders
  float fx = (u + 0.5f) * ((float)w / S) - 0.5f, fy = (v_ + 0.5f) * ((float)h / S) - 0.5f;
  fx = fminf(fmaxf(fx, 0.f), (float)(w - 1)); fy = fminf(fmaxf(fy, 0.f), (float)(h - 1));
  const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
  const float ax = fx - x0, ay = fy - y0;
  float a[3], b[3], c[3], d[3];
  aug_pixel(it, mean_grey, x0, y0, a); aug_pixel(it, mean_grey, x1, y0, b);
  aug_pixel(it, mean_grey, x0, y1, c); aug_pixel(it, mean_grey, x1, y1, d);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float val = (1.f - ay) * ((1.f - ax) * a[ch] + ax * b[ch]) + ay * ((1.f - ax) * c[ch] + ax * d[ch]);
    if (mean) val = (val - mean[ch]) / stdv[ch];
    it.out[(long long)ch * S * S + p] = val;
  }
}

}  // namespace

static_assert(sizeof(AugItem) == sizeof(cape_augment_item), "cape_augment_item layout");

extern "C" int cape_augment_batch(const cape_augment_item* items_dev, int n_items, int max_pixels, int out_size, const float* mean,
                                  const float* stdv, cape_stream_t stream) {
  CAPE_REQUIRE(items_dev != nullptr && n_items >= 0 && out_size >= 1 && max_pixels >= 1, "cape_augment_batch: bad arguments");
  CAPE_REQUIRE((mean == nullptr) == (stdv == nullptr), "cape_augment_batch: mean and std come together");
  if (n_items == 0) return 0;
  CAPE_REQUIRE(n_items <= 65535, "cape_augment_batch: at most 65535 images per launch");
  int bx = (max_pixels + 255) / 256;
  if (bx > 1024) bx = 1024;                                  // grid-stride over the larger crops
  const AugItem* it = reinterpret_cast<const AugItem*>(items_dev);
  hipLaunchKernelGGL(augment_warp_kernel, dim3((unsigned)bx, (unsigned)n_items), dim3(256), 0, as_stream(stream), it);
  hipLaunchKernelGGL(augment_finish_kernel, dim3((unsigned)((out_size * out_size + 255) / 256), (unsigned)n_items), dim3(256), 0,
                     as_stream(stream), it, out_size, mean, stdv);
  CAPE_LAUNCH_CHECK("cape_augment_batch");
  return 0;
}
