// elementwise.hip -- HBM-bound pointwise / small-row kernels of the CAPE path (gfx950).
// Grid-stride loops, 16-byte accesses where the layout allows; transcendental tables
// (the 128-entry `dim_t` temperature table) come from the host so the values match torch bit for bit.
#include "common.h"

namespace {

constexpr int TPB = 256;
inline unsigned grid_for(long long n, int per = 1) {
  long long b = (n + (long long)TPB * per - 1) / ((long long)TPB * per);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
#define GSTRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__global__ void add_kernel(const float* a, const float* b, float* o, long long n) {
  const long long n4 = n >> 2;
  GSTRIDE(i, n4) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  GSTRIDE(i, n & 3) o[(n4 << 2) + i] = a[(n4 << 2) + i] + b[(n4 << 2) + i];
}

// o = sum of up to 8 equally shaped tensors (gradient fan-in of a tensor with several consumers): one pass, k reads + 1 write
struct AddNP { const float* src[8]; int k; };
__global__ void add_n_kernel(const AddNP p, float* o, long long n) {
  const long long n4 = n >> 2;
  GSTRIDE(i, n4) {
    float4 s = reinterpret_cast<const float4*>(p.src[0])[i];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < p.k) {
        const float4 y = reinterpret_cast<const float4*>(p.src[j])[i];
        s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w;
      }
    reinterpret_cast<float4*>(o)[i] = s;
  }
  GSTRIDE(i, n & 3) {
    float s = p.src[0][(n4 << 2) + i];
    for (int j = 1; j < p.k; ++j) s += p.src[j][(n4 << 2) + i];
    o[(n4 << 2) + i] = s;
  }
}

// the same with row-strided sources: source j is (rows, cols) with row stride ld[j] (a column block of a wider buffer, e.g. the
// q part of the decoder's wide [dq | dk | dv] gradient), cols % 4 == 0
struct AddNRowsP { const float* src[8]; long long ld[8]; int k; };
__global__ void add_n_rows_kernel(const AddNRowsP p, float* o, long long ldo, long long rows, int cols) {
  const int c4 = cols >> 2;
  GSTRIDE(i, rows * c4) {
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    float4 s = *reinterpret_cast<const float4*>(p.src[0] + r * p.ld[0] + c);
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < p.k) {
        const float4 y = *reinterpret_cast<const float4*>(p.src[j] + r * p.ld[j] + c);
        s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w;
      }
    *reinterpret_cast<float4*>(o + r * ldo + c) = s;      // (the output may be one of the sources: element-wise read, then write)
  }
}

struct Cls4 { const float* c[4]; };
__global__ void interleave2x2_kernel(const Cls4 cls, const float* acc, float* __restrict__ out, int N, int H, int W, int C) {
  const int c4n = C >> 2, H2 = H >> 1, W2 = W >> 1;
  const long long tot = (long long)N * H * W * c4n;
  GSTRIDE(i, tot) {
    const int c = (int)(i % c4n) * 4;
    long long pix = i / c4n;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    const float* src = cls.c[(y & 1) * 2 + (x & 1)];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src) v = *reinterpret_cast<const float4*>(src + (((long long)n * H2 + (y >> 1)) * W2 + (x >> 1)) * C + c);
    const long long o = (((long long)n * H + y) * W + x) * C + c;
    if (acc) { const float4 a = *reinterpret_cast<const float4*>(acc + o); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    *reinterpret_cast<float4*>(out + o) = v;
  }
}

__global__ void nchw_to_nhwc_kernel(const float* x, float* o, int N, int C, int H, int W, int Cp) {
  const long long tot = (long long)N * H * W * Cp;
  GSTRIDE(i, tot) {
    const int c = (int)(i % Cp);
    const long long pix = i / Cp;
    const int w = (int)(pix % W);
    const long long t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    o[i] = c < C ? x[(((long long)n * C + c) * H + h) * W + w] : 0.f;
  }
}

__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float eps, float* sc,
                               float* sh, int C) {
  GSTRIDE(i, C) {
    const float s = w[i] * rsqrtf(rv[i] + eps);
    sc[i] = s;
    sh[i] = b[i] - rm[i] * s;
  }
}

__global__ void maxpool_kernel(const float* x, float* o, int N, int H, int W, int C, int OH, int OW) {
  const int C4 = C >> 2;
  const long long tot = (long long)N * OH * OW * C4;
  GSTRIDE(i, tot) {
    const int c = (int)(i % C4) * 4;
    const long long pix = i / C4;
    const int ox = (int)(pix % OW);
    const long long t = pix / OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long long)n * H + iy) * W + ix) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(o + pix * C + c) = m;
  }
}

__global__ void bn_relu_bwd_kernel(const float* dy, const float* y, const float* scale, float* d_pre, float* d_res,
                                   long long rows, int C, int relu) {
  const int C4 = C >> 2;
  const long long tot = rows * C4;
  GSTRIDE(i, tot) {
    const int c = (int)(i % C4) * 4;
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    if (relu) {
      const float4 v = reinterpret_cast<const float4*>(y)[i];
      g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f;
      g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
    }
    if (d_res) reinterpret_cast<float4*>(d_res)[i] = g;
    if (scale) {
      const float4 s = *reinterpret_cast<const float4*>(scale + c);
      g.x *= s.x; g.y *= s.y; g.z *= s.z; g.w *= s.w;
    }
    reinterpret_cast<float4*>(d_pre)[i] = g;
  }
}

// in place: x = [relu]( x * scale[c] + bias[c] [+ residual] ) -- the FrozenBN / ReLU / shortcut epilogue of a convolution whose
// contraction was split over k (atomic partial sums cannot carry an epilogue)
__global__ void affine_act_kernel(float* x, const float* scale, const float* bias, const float* residual, long long rows, int C,
                                  int relu) {
  const int C4 = C >> 2;
  const long long tot = rows * C4;
  GSTRIDE(i, tot) {
    const int c = (int)(i % C4) * 4;
    float4 v = reinterpret_cast<float4*>(x)[i];
    if (scale) { const float4 s = *reinterpret_cast<const float4*>(scale + c); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (residual) { const float4 r = reinterpret_cast<const float4*>(residual)[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

__global__ void relu_drop_bwd_kernel(const float* dh, const float* h, float* d_pre, long long n, float inv_keep) {
  GSTRIDE(i, n) d_pre[i] = h[i] > 0.f ? dh[i] * inv_keep : 0.f;
}

// image sine position embedding (+ level embedding); thread = (n, pixel, channel)
__global__ void pos_sine_level_kernel(const uint8_t* mask, const float* lvl, const float* dim_t, float* out,
                                      long long image_stride, int N, int h, int w, int C) {
  const int half = C >> 1;
  const long long tot = (long long)N * h * w * C;
  GSTRIDE(i, tot) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int x = (int)(pix % w);
    const long long t = pix / w;
    const int y = (int)(t % h);
    const int n = (int)(t / h);
    const uint8_t* m = mask + (long long)n * h * w;
    float e, tot_e;
    if (c < half) {   // pos_y block: cumsum over rows of this column
      int cs = 0, all = 0;
      for (int r = 0; r < h; ++r) { const int v = m[r * w + x] == 0; all += v; if (r <= y) cs += v; }
      e = (float)cs; tot_e = (float)all;
    } else {
      int cs = 0, all = 0;
      for (int q = 0; q < w; ++q) { const int v = m[y * w + q] == 0; all += v; if (q <= x) cs += v; }
      e = (float)cs; tot_e = (float)all;
    }
    const int k = c < half ? c : c - half;
    const float v = (e - 0.5f) / (tot_e + 1e-6f) * 6.283185307179586f / dim_t[k];
    const float r = (k & 1) ? cosf(v) : sinf(v);
    out[(long long)n * image_stride + ((long long)y * w + x) * C + c] = r + lvl[c];
  }
}

// ---- token embedding -------------------------------------------------------------------------
__global__ void token_embed_fwd_kernel(const float* table, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                                       const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                                       const float* dy2, float* out, long long R, int C) {
  const int C4 = C >> 2;
  const long long tot = R * C4;
  GSTRIDE(i, tot) {
    const long long r = i / C4;
    const int c = (int)(i % C4) * 4;
    const float w11 = dx2[r] * dy2[r], w21 = dx1[r] * dy2[r], w12 = dx2[r] * dy1[r], w22 = dx1[r] * dy1[r];
    const float4 a = *reinterpret_cast<const float4*>(table + s11[r] * C + c);
    const float4 b = *reinterpret_cast<const float4*>(table + s21[r] * C + c);
    const float4 d = *reinterpret_cast<const float4*>(table + s12[r] * C + c);
    const float4 e = *reinterpret_cast<const float4*>(table + s22[r] * C + c);
    // same association order as the reference expression (e11*dx2*dy2 + e21*dx1*dy2 + e12*dx2*dy1 + e22*dx1*dy1)
    float4 o;
    o.x = a.x * dx2[r] * dy2[r] + b.x * dx1[r] * dy2[r] + d.x * dx2[r] * dy1[r] + e.x * dx1[r] * dy1[r];
    o.y = a.y * dx2[r] * dy2[r] + b.y * dx1[r] * dy2[r] + d.y * dx2[r] * dy1[r] + e.y * dx1[r] * dy1[r];
    o.z = a.z * dx2[r] * dy2[r] + b.z * dx1[r] * dy2[r] + d.z * dx2[r] * dy1[r] + e.z * dx1[r] * dy1[r];
    o.w = a.w * dx2[r] * dy2[r] + b.w * dx1[r] * dy2[r] + d.w * dx2[r] * dy1[r] + e.w * dx1[r] * dy1[r];
    (void)w11; (void)w21; (void)w12; (void)w22;
    *reinterpret_cast<float4*>(out + r * C + c) = o;
  }
}

__global__ void token_embed_bwd_kernel(const float* d_out, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                                       const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                                       const float* dy2, float* d_table, long long R, int C, int pad_idx) {
  const long long tot = R * C;
  GSTRIDE(i, tot) {
    const long long r = i / C;
    const int c = (int)(i % C);
    const float g = d_out[i];
    const int64_t i11 = s11[r], i21 = s21[r], i12 = s12[r], i22 = s22[r];
    if (i11 != pad_idx) atomicAdd(d_table + i11 * C + c, g * dx2[r] * dy2[r]);
    if (i21 != pad_idx) atomicAdd(d_table + i21 * C + c, g * dx1[r] * dy2[r]);
    if (i12 != pad_idx) atomicAdd(d_table + i12 * C + c, g * dx2[r] * dy1[r]);
    if (i22 != pad_idx) atomicAdd(d_table + i22 * C + c, g * dx1[r] * dy1[r]);
  }
}

// ---- decoder query sine embedding ---------------------------------------------------------------
__global__ void query_sine_fwd_kernel(const float* ref, const float* dim_t, float* out, long long R) {
  const long long tot = R * 256;
  GSTRIDE(i, tot) {
    const long long r = i >> 8;
    const int c = (int)(i & 255);
    const int a = c >> 7, k = c & 127;
    const float v = ref[r * 2 + a] * 6.283185307179586f / dim_t[k];
    out[i] = (k & 1) ? cosf(v) : sinf(v);
  }
}

// one wave per row (4 rows per block, no barrier, no LDS): lane l owns channels l, l + 64 (x axis) and l + 128, l + 192 (y axis)
__global__ void __launch_bounds__(256) query_sine_bwd_kernel(const float* d_out, const float* ref, const float* dim_t,
                                                              float* d_ref, int accumulate, long long R) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;                                            // wave-uniform
  const float rx = ref[r * 2], ry = ref[r * 2 + 1];
  float gx = 0.f, gy = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;                                 // channel inside an axis: 0..127
    const float dt = dim_t[k];
    const float sc = 6.283185307179586f / dt;
    const float vx = rx * 6.283185307179586f / dt, vy = ry * 6.283185307179586f / dt;
    gx += d_out[r * 256 + k] * ((k & 1) ? -sinf(vx) : cosf(vx)) * sc;
    gy += d_out[r * 256 + 128 + k] * ((k & 1) ? -sinf(vy) : cosf(vy)) * sc;
  }
  gx = wave_sum(gx);
  gy = wave_sum(gy);
  if (lane == 0) {
    if (accumulate) { d_ref[r * 2] += gx; d_ref[r * 2 + 1] += gy; }
    else { d_ref[r * 2] = gx; d_ref[r * 2 + 1] = gy; }
  }
}

// ---- refinement / sigmoid -----------------------------------------------------------------------
__device__ __forceinline__ float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  return logf(x1 / x2);
}
__global__ void refine_fwd_kernel(const float* delta, const float* ref, float* out, long long n) {
  GSTRIDE(i, n) {
    const float z = delta[i] + inv_sigmoid(ref[i]);
    out[i] = 1.f / (1.f + expf(-z));
  }
}
__global__ void refine_bwd_kernel(const float* d_new, const float* new_ref, const float* ref, float* d_delta, float* d_ref,
                                  int acc, long long n) {
  GSTRIDE(i, n) {
    const float s = new_ref[i];
    const float dz = d_new[i] * s * (1.f - s);
    d_delta[i] = dz;
    if (d_ref) {
      const float x = ref[i];
      float dx = 0.f;
      if (x >= 0.f && x <= 1.f) {
        if (x >= 1e-5f) dx += 1.f / x;                 // d log(max(x, eps))
        if (1.f - x >= 1e-5f) dx += 1.f / (1.f - x);   // d -log(max(1-x, eps))
      }
      const float g = dz * dx;
      if (acc) d_ref[i] += g; else d_ref[i] = g;
    }
  }
}
__global__ void sigmoid_fwd_kernel(const float* x, float* y, long long n) {
  GSTRIDE(i, n) y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* dy, const float* y, float* dx, int acc, long long n) {
  GSTRIDE(i, n) {
    const float g = dy[i] * y[i] * (1.f - y[i]);
    if (acc) dx[i] += g; else dx[i] = g;
  }
}
__global__ void ref_scale_fwd_kernel(const float* ref, const float* vr, float* out, long long R, int rpi, int L) {
  const long long tot = R * L * 2;
  GSTRIDE(i, tot) {
    const int a = (int)(i & 1);
    const long long t = i >> 1;
    const int l = (int)(t % L);
    const long long r = t / L;
    const long long n = r / rpi;
    out[i] = ref[r * 2 + a] * vr[(n * L + l) * 2 + a];
  }
}
__global__ void ref_scale_bwd_kernel(const float* d_in, const float* vr, float* d_ref, int acc, long long R, int rpi, int L) {
  const long long tot = R * 2;
  GSTRIDE(i, tot) {
    const int a = (int)(i & 1);
    const long long r = i >> 1;
    const long long n = r / rpi;
    float g = 0.f;
    for (int l = 0; l < L; ++l) g += d_in[(r * L + l) * 2 + a] * vr[(n * L + l) * 2 + a];
    if (acc) d_ref[i] += g; else d_ref[i] = g;
  }
}
// exact (erf) GELU, nn.GELU() default
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
  GSTRIDE(i, n) { const float v = x[i]; out[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
}
// out[r][c] = x[r][c] + y[r][c] * gamma[c]   (gamma NULL = 1): residual add behind a LayerScale
__global__ void scale_residual_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                                      float* __restrict__ out, long long rows, int C) {
  const long long tot = rows * C;
  GSTRIDE(i, tot) out[i] = x[i] + y[i] * (gamma ? gamma[i % C] : 1.f);
}
// d/dx of the exact GELU: Phi(x) + x * phi(x)
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx, long long n) {
  GSTRIDE(i, n) {
    const float v = x[i];
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
    dx[i] = g[i] * (cdf + v * pdf);
  }
}
// backward of out = x + gamma * y with respect to y and gamma: dy = gamma * g ; dgamma[c] += sum_r g[r][c] * y[r][c].
// One column per lane (C <= 1024 columns over grid.x), row slabs over grid.y, one atomic per column and block.
__global__ void __launch_bounds__(256) scale_residual_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                  const float* __restrict__ gamma, float* __restrict__ dy,
                                                                  float* __restrict__ dgamma, long long rows, int C,
                                                                  long long rows_per_block) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (col < C) {
    const float gm = gamma[col];
    for (long long r = r0 + w; r < r1; r += 4) {
      const float gv = g[r * C + col];
      s += gv * y[r * C + col];
      dy[r * C + col] = gm * gv;
    }
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < C) atomicAdd(&dgamma[col], part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}
__global__ void zero_rows_kernel(float* x, const uint8_t* rowmask, long long rows, int C) {
  const long long tot = rows * C;
  GSTRIDE(i, tot) if (rowmask[i / C]) x[i] = 0.f;
}

}  // namespace

#define LAUNCH1(kern, n, per, ...)                                                               \
  hipLaunchKernelGGL(kern, dim3(grid_for((n), (per))), dim3(TPB), 0, as_stream(stream), __VA_ARGS__)

extern "C" int cape_add_f32(const float* a, const float* b, float* out, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(a && b && out && n >= 0, "cape_add_f32: bad arguments");
  if (n == 0) return 0;
  CAPE_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "cape_add_f32: pointers must be 16-byte aligned");
  LAUNCH1(add_kernel, n / 4 + 1, 1, a, b, out, n);
  CAPE_LAUNCH_CHECK("cape_add_f32");
  return 0;
}

namespace {
struct LevelStarts { int start[4]; int L; };
__global__ void __launch_bounds__(256) level_embed_add_kernel(const float* __restrict__ base, const float* __restrict__ level_embed,
                                                              float* __restrict__ out, LevelStarts ls, long long rows, int S, int C4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one float4 per thread
  if (i >= rows * C4) return;
  const long long row = i / C4;
  const int c4 = (int)(i - row * C4), s = (int)(row % S);
  int l = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) l += (k < ls.L && s >= ls.start[k]) ? 1 : 0;
  const float4 a = reinterpret_cast<const float4*>(base)[i];
  const float4 e = reinterpret_cast<const float4*>(level_embed)[l * C4 + c4];
  reinterpret_cast<float4*>(out)[i] = make_float4(a.x + e.x, a.y + e.y, a.z + e.z, a.w + e.w);
}
}  // namespace

extern "C" int cape_level_embed_add(const float* base, const float* level_embed, const int* level_start, float* out, int N, int S, int L,
                                    int C, cape_stream_t stream) {
  CAPE_REQUIRE(base && level_embed && level_start && out, "cape_level_embed_add: null pointer");
  CAPE_REQUIRE(L >= 1 && L <= 4 && C > 0 && C % 4 == 0 && S > 0, "cape_level_embed_add: L=%d C=%d S=%d", L, C, S);
  if (N <= 0) return 0;
  CAPE_REQUIRE(((reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(level_embed) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "cape_level_embed_add: pointers must be 16-byte aligned");
  LevelStarts ls;
  ls.L = L;
  for (int k = 0; k < 4; ++k) ls.start[k] = k < L ? level_start[k] : S;
  CAPE_REQUIRE(ls.start[0] == 0, "cape_level_embed_add: level 0 starts at 0");
  const long long rows = (long long)N * S, n4 = rows * (C / 4);
  CAPE_REQUIRE((n4 + 255) / 256 < (1ll << 31), "cape_level_embed_add: too large");
  hipLaunchKernelGGL(level_embed_add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), base, level_embed, out, ls, rows,
                     S, C / 4);
  CAPE_LAUNCH_CHECK("cape_level_embed_add");
  return 0;
}

extern "C" int cape_add_n_f32(const float* const* srcs, int k, float* out, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(srcs && out && k >= 1 && k <= 8 && n >= 0, "cape_add_n_f32: 1..8 sources");
  if (n == 0) return 0;
  AddNP p;
  uintptr_t al = reinterpret_cast<uintptr_t>(out);
  for (int j = 0; j < 8; ++j) {
    p.src[j] = j < k ? srcs[j] : srcs[0];
    CAPE_REQUIRE(p.src[j] != nullptr, "cape_add_n_f32: null source");
    al |= reinterpret_cast<uintptr_t>(p.src[j]);
  }
  CAPE_REQUIRE((al & 15) == 0, "cape_add_n_f32: pointers must be 16-byte aligned");
  p.k = k;
  LAUNCH1(add_n_kernel, n / 4 + 1, 1, p, out, n);
  CAPE_LAUNCH_CHECK("cape_add_n_f32");
  return 0;
}

extern "C" int cape_add_n_rows_f32(const float* const* srcs, const long long* lds, int k, float* out, long long ldo, long long rows,
                                   int cols, cape_stream_t stream) {
  CAPE_REQUIRE(srcs && lds && out && k >= 1 && k <= 8 && rows >= 0 && cols >= 4 && cols % 4 == 0, "cape_add_n_rows_f32: 1..8 sources, cols % 4 == 0");
  if (rows == 0) return 0;
  AddNRowsP p;
  uintptr_t al = reinterpret_cast<uintptr_t>(out);
  for (int j = 0; j < 8; ++j) {
    p.src[j] = j < k ? srcs[j] : srcs[0];
    p.ld[j] = j < k ? lds[j] : lds[0];
    CAPE_REQUIRE(p.src[j] != nullptr && p.ld[j] >= cols && p.ld[j] % 4 == 0, "cape_add_n_rows_f32: null source or bad row stride");
    al |= reinterpret_cast<uintptr_t>(p.src[j]);
  }
  CAPE_REQUIRE((al & 15) == 0, "cape_add_n_rows_f32: pointers must be 16-byte aligned");
  CAPE_REQUIRE(ldo >= cols && ldo % 4 == 0, "cape_add_n_rows_f32: bad output row stride");
  p.k = k;
  LAUNCH1(add_n_rows_kernel, rows * (cols / 4), 1, p, out, ldo, rows, cols);
  CAPE_LAUNCH_CHECK("cape_add_n_rows_f32");
  return 0;
}

extern "C" int cape_interleave2x2_f32(const float* const* cls, const float* acc, float* out, int N, int H, int W, int C, cape_stream_t stream) {
  CAPE_REQUIRE(cls && out && N >= 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0 && (C % 4) == 0,
               "cape_interleave2x2_f32: even H, W and C %% 4 == 0");
  if (N == 0) return 0;
  Cls4 c4;
  uintptr_t al = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(acc);
  for (int k = 0; k < 4; ++k) { c4.c[k] = cls[k]; al |= reinterpret_cast<uintptr_t>(cls[k]); }
  CAPE_REQUIRE((al & 15) == 0, "cape_interleave2x2_f32: pointers must be 16-byte aligned");
  LAUNCH1(interleave2x2_kernel, (long long)N * H * W * (C / 4), 1, c4, acc, out, N, H, W, C);
  CAPE_LAUNCH_CHECK("cape_interleave2x2_f32");
  return 0;
}

extern "C" int cape_gelu_f32(const float* x, float* out, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(x && out && n >= 0, "cape_gelu_f32: bad arguments");
  if (n == 0) return 0;
  LAUNCH1(gelu_kernel, n, 4, x, out, n);
  CAPE_LAUNCH_CHECK("cape_gelu_f32");
  return 0;
}

extern "C" int cape_gelu_bwd_f32(const float* x, const float* g, float* dx, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(x && g && dx && n >= 0, "cape_gelu_bwd_f32: bad arguments");
  if (n == 0) return 0;
  LAUNCH1(gelu_bwd_kernel, n, 4, x, g, dx, n);
  CAPE_LAUNCH_CHECK("cape_gelu_bwd_f32");
  return 0;
}

extern "C" int cape_scale_residual_bwd_f32(const float* g, const float* y, const float* gamma, float* dy, float* dgamma,
                                           long long rows, int C, cape_stream_t stream) {
  CAPE_REQUIRE(g && y && gamma && dy && dgamma && rows >= 0 && C > 0, "cape_scale_residual_bwd_f32: bad arguments");
  if (rows == 0) return 0;
  long long slabs = (rows + 63) / 64;
  if (slabs > 256) slabs = 256;
  const long long rpb = (rows + slabs - 1) / slabs;
  hipLaunchKernelGGL(scale_residual_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0,
                     as_stream(stream), g, y, gamma, dy, dgamma, rows, C, rpb);
  CAPE_LAUNCH_CHECK("cape_scale_residual_bwd_f32");
  return 0;
}

extern "C" int cape_scale_residual_f32(const float* x, const float* y, const float* gamma, float* out, long long rows, int C,
                                       cape_stream_t stream) {
  CAPE_REQUIRE(x && y && out && rows >= 0 && C > 0, "cape_scale_residual_f32: bad arguments");
  if (rows == 0) return 0;
  LAUNCH1(scale_residual_kernel, rows * C, 4, x, y, gamma, out, rows, C);
  CAPE_LAUNCH_CHECK("cape_scale_residual_f32");
  return 0;
}

extern "C" int cape_nchw_to_nhwc(const float* x, float* out, int N, int C, int H, int W, int Cp, cape_stream_t stream) {
  CAPE_REQUIRE(x && out && Cp >= C && C > 0, "cape_nchw_to_nhwc: bad arguments");
  const long long n = (long long)N * H * W * Cp;
  if (n == 0) return 0;
  LAUNCH1(nchw_to_nhwc_kernel, n, 1, x, out, N, C, H, W, Cp);
  CAPE_LAUNCH_CHECK("cape_nchw_to_nhwc");
  return 0;
}

extern "C" int cape_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps, float* scale,
                            float* shift, int C, cape_stream_t stream) {
  CAPE_REQUIRE(w && b && rm && rv && scale && shift && C > 0, "cape_bn_fold: bad arguments");
  LAUNCH1(bn_fold_kernel, C, 1, w, b, rm, rv, eps, scale, shift, C);
  CAPE_LAUNCH_CHECK("cape_bn_fold");
  return 0;
}

extern "C" int cape_maxpool3x3s2_nhwc(const float* x, float* out, int N, int H, int W, int C, cape_stream_t stream) {
  CAPE_REQUIRE(x && out && (C % 4) == 0, "cape_maxpool3x3s2_nhwc: bad arguments (C %% 4 != 0?)");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)N * OH * OW * (C / 4);
  if (n == 0) return 0;
  LAUNCH1(maxpool_kernel, n, 1, x, out, N, H, W, C, OH, OW);
  CAPE_LAUNCH_CHECK("cape_maxpool3x3s2_nhwc");
  return 0;
}

extern "C" int cape_bn_relu_bwd(const float* dy, const float* y, const float* scale, float* d_pre, float* d_res,
                                long long rows, int C, int relu, cape_stream_t stream) {
  CAPE_REQUIRE(dy && d_pre && (C % 4) == 0 && (!relu || y), "cape_bn_relu_bwd: bad arguments");
  const long long n = rows * (C / 4);
  if (n == 0) return 0;
  LAUNCH1(bn_relu_bwd_kernel, n, 1, dy, y, scale, d_pre, d_res, rows, C, relu);
  CAPE_LAUNCH_CHECK("cape_bn_relu_bwd");
  return 0;
}

extern "C" int cape_affine_act_f32(float* x, const float* scale, const float* bias, const float* residual, long long rows, int C,
                                   int relu, cape_stream_t stream) {
  CAPE_REQUIRE(x && rows >= 0 && C > 0 && C % 4 == 0, "cape_affine_act_f32: bad arguments (C must be a multiple of 4)");
  CAPE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) |
                 reinterpret_cast<uintptr_t>(residual)) & 15) == 0, "cape_affine_act_f32: pointers must be 16-byte aligned");
  const long long n = rows * (C / 4);
  if (n == 0) return 0;
  LAUNCH1(affine_act_kernel, n, 1, x, scale, bias, residual, rows, C, relu);
  CAPE_LAUNCH_CHECK("cape_affine_act_f32");
  return 0;
}

extern "C" int cape_relu_drop_bwd(const float* dh, const float* h, float* d_pre, long long n, float inv_keep,
                                  cape_stream_t stream) {
  CAPE_REQUIRE(dh && h && d_pre, "cape_relu_drop_bwd: null pointer");
  if (n == 0) return 0;
  LAUNCH1(relu_drop_bwd_kernel, n, 4, dh, h, d_pre, n, inv_keep);
  CAPE_LAUNCH_CHECK("cape_relu_drop_bwd");
  return 0;
}

extern "C" int cape_pos_sine_level(const uint8_t* mask, const float* level_embed_l, const float* dim_t, float* out,
                                   long long out_image_stride, int N, int h, int w, int C, cape_stream_t stream) {
  CAPE_REQUIRE(mask && level_embed_l && dim_t && out && C == 256, "cape_pos_sine_level: bad arguments (C must be 256)");
  const long long n = (long long)N * h * w * C;
  if (n == 0) return 0;
  LAUNCH1(pos_sine_level_kernel, n, 1, mask, level_embed_l, dim_t, out, out_image_stride, N, h, w, C);
  CAPE_LAUNCH_CHECK("cape_pos_sine_level");
  return 0;
}

extern "C" int cape_token_embed_fwd(const float* table, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                                    const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                                    const float* dy2, float* out, long long R, int C, int vocab, cape_stream_t stream) {
  CAPE_REQUIRE(table && s11 && s21 && s12 && s22 && dx1 && dx2 && dy1 && dy2 && out && (C % 4) == 0 && vocab > 0,
               "cape_token_embed_fwd: bad arguments");
  if (R == 0) return 0;
  LAUNCH1(token_embed_fwd_kernel, R * (C / 4), 1, table, s11, s21, s12, s22, dx1, dx2, dy1, dy2, out, R, C);
  CAPE_LAUNCH_CHECK("cape_token_embed_fwd");
  return 0;
}

extern "C" int cape_token_embed_bwd(const float* d_out, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                                    const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                                    const float* dy2, float* d_table, long long R, int C, int vocab, int pad_idx,
                                    cape_stream_t stream) {
  CAPE_REQUIRE(d_out && s11 && s21 && s12 && s22 && dx1 && dx2 && dy1 && dy2 && d_table && vocab > 0,
               "cape_token_embed_bwd: bad arguments");
  if (R == 0) return 0;
  LAUNCH1(token_embed_bwd_kernel, R * C, 1, d_out, s11, s21, s12, s22, dx1, dx2, dy1, dy2, d_table, R, C, pad_idx);
  CAPE_LAUNCH_CHECK("cape_token_embed_bwd");
  return 0;
}

extern "C" int cape_query_sine_fwd(const float* ref, const float* dim_t, float* out, long long R, cape_stream_t stream) {
  CAPE_REQUIRE(ref && dim_t && out, "cape_query_sine_fwd: null pointer");
  if (R == 0) return 0;
  LAUNCH1(query_sine_fwd_kernel, R * 256, 1, ref, dim_t, out, R);
  CAPE_LAUNCH_CHECK("cape_query_sine_fwd");
  return 0;
}

extern "C" int cape_query_sine_bwd(const float* d_out, const float* ref, const float* dim_t, float* d_ref, int accumulate,
                                   long long R, cape_stream_t stream) {
  CAPE_REQUIRE(d_out && ref && dim_t && d_ref && R < (1ll << 31), "cape_query_sine_bwd: bad arguments");
  if (R == 0) return 0;
  hipLaunchKernelGGL(query_sine_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, as_stream(stream), d_out, ref, dim_t, d_ref,
                     accumulate, R);
  CAPE_LAUNCH_CHECK("cape_query_sine_bwd");
  return 0;
}

extern "C" int cape_refine_fwd(const float* delta, const float* ref, float* new_ref, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(delta && ref && new_ref, "cape_refine_fwd: null pointer");
  if (n == 0) return 0;
  LAUNCH1(refine_fwd_kernel, n, 1, delta, ref, new_ref, n);
  CAPE_LAUNCH_CHECK("cape_refine_fwd");
  return 0;
}

extern "C" int cape_refine_bwd(const float* d_new, const float* new_ref, const float* ref, float* d_delta, float* d_ref,
                               int accumulate_ref, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(d_new && new_ref && ref && d_delta, "cape_refine_bwd: null pointer");
  if (n == 0) return 0;
  LAUNCH1(refine_bwd_kernel, n, 1, d_new, new_ref, ref, d_delta, d_ref, accumulate_ref, n);
  CAPE_LAUNCH_CHECK("cape_refine_bwd");
  return 0;
}

extern "C" int cape_sigmoid_fwd(const float* x, float* y, long long n, cape_stream_t stream) {
  CAPE_REQUIRE(x && y, "cape_sigmoid_fwd: null pointer");
  if (n == 0) return 0;
  LAUNCH1(sigmoid_fwd_kernel, n, 1, x, y, n);
  CAPE_LAUNCH_CHECK("cape_sigmoid_fwd");
  return 0;
}

extern "C" int cape_sigmoid_bwd(const float* dy, const float* y, float* dx, int accumulate, long long n,
                                cape_stream_t stream) {
  CAPE_REQUIRE(dy && y && dx, "cape_sigmoid_bwd: null pointer");
  if (n == 0) return 0;
  LAUNCH1(sigmoid_bwd_kernel, n, 1, dy, y, dx, accumulate, n);
  CAPE_LAUNCH_CHECK("cape_sigmoid_bwd");
  return 0;
}

extern "C" int cape_ref_scale_fwd(const float* ref, const float* valid_ratios, float* ref_in, long long R,
                                  int rows_per_image, int L, cape_stream_t stream) {
  CAPE_REQUIRE(ref && valid_ratios && ref_in && rows_per_image > 0 && L > 0, "cape_ref_scale_fwd: bad arguments");
  if (R == 0) return 0;
  LAUNCH1(ref_scale_fwd_kernel, R * L * 2, 1, ref, valid_ratios, ref_in, R, rows_per_image, L);
  CAPE_LAUNCH_CHECK("cape_ref_scale_fwd");
  return 0;
}

extern "C" int cape_ref_scale_bwd(const float* d_ref_in, const float* valid_ratios, float* d_ref, int accumulate,
                                  long long R, int rows_per_image, int L, cape_stream_t stream) {
  CAPE_REQUIRE(d_ref_in && valid_ratios && d_ref && rows_per_image > 0 && L > 0, "cape_ref_scale_bwd: bad arguments");
  if (R == 0) return 0;
  LAUNCH1(ref_scale_bwd_kernel, R * 2, 1, d_ref_in, valid_ratios, d_ref, accumulate, R, rows_per_image, L);
  CAPE_LAUNCH_CHECK("cape_ref_scale_bwd");
  return 0;
}

extern "C" int cape_zero_rows(float* x, const uint8_t* rowmask, long long rows, int C, cape_stream_t stream) {
  CAPE_REQUIRE(x && rowmask && C > 0, "cape_zero_rows: bad arguments");
  if (rows == 0) return 0;
  LAUNCH1(zero_rows_kernel, rows * C, 1, x, rowmask, rows, C);
  CAPE_LAUNCH_CHECK("cape_zero_rows");
  return 0;
}
