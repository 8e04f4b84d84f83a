// norm.hip -- fused (residual + dropout + LayerNorm) and GroupNorm, forward and backward.
// HBM-bound row kernels: one wave64 per row, 16-byte loads, wave shuffles for the row statistics.
#include "common.h"

namespace {

constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 1024

// ---------------------------------------------------------------------------------------------
// forward: out = LN(x + drop(y)); optional out_pos = out + pos
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ out, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, const float* __restrict__ pos,
                                                          float* __restrict__ out_pos, int rows, int C, uint32_t thresh,
                                                          float inv_keep, const uint64_t* rng_state, uint32_t rng_stream) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = C >> 8;             // full float4 rounds of 64 lanes
  const int rem = (C & 255) >> 2;    // lanes active in the last partial round
  uint64_t seed = 0, step = 0;
  if (thresh) { seed = rng_state[0]; step = rng_state[1]; }
  float4 s[LN_MAXV];
  float sum = 0.f;
  const long long base = (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool act = (i < nv) || (i == nv && lane < rem);
    s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      const int c = (i * 64 + lane) * 4;
      float4 v = *reinterpret_cast<const float4*>(x + base + c);
      if (y) {
        float4 w = *reinterpret_cast<const float4*>(y + base + c);
        if (thresh) {
          const uint64_t idx = (uint64_t)base + c;
          w.x = cape_keep(seed, step, rng_stream, idx + 0, thresh) ? w.x * inv_keep : 0.f;
          w.y = cape_keep(seed, step, rng_stream, idx + 1, thresh) ? w.y * inv_keep : 0.f;
          w.z = cape_keep(seed, step, rng_stream, idx + 2, thresh) ? w.z * inv_keep : 0.f;
          w.w = cape_keep(seed, step, rng_stream, idx + 3, thresh) ? w.w * inv_keep : 0.f;
        }
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      s[i] = v;
      sum += v.x + v.y + v.z + v.w;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool act = (i < nv) || (i == nv && lane < rem);
    if (act) {
      const float a = s[i].x - mean, b = s[i].y - mean, c = s[i].z - mean, d = s[i].w - mean;
      sq += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + 1e-5f);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool act = (i < nv) || (i == nv && lane < rem);
    if (act) {
      const int c = (i * 64 + lane) * 4;
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (s[i].x - mean) * rstd * g.x + b.x;
      o.y = (s[i].y - mean) * rstd * g.y + b.y;
      o.z = (s[i].z - mean) * rstd * g.z + b.z;
      o.w = (s[i].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(out + base + c) = o;
      if (out_pos) {
        const float4 pp = *reinterpret_cast<const float4*>(pos + base + c);
        o.x += pp.x; o.y += pp.y; o.z += pp.z; o.w += pp.w;
        *reinterpret_cast<float4*>(out_pos + base + c) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  Each block walks ROWS_PER_BLOCK rows (4 waves x 8 rows), keeps dgamma/dbeta partials of
// its columns in registers, combines the 4 waves through LDS and issues one atomic per column.
// ---------------------------------------------------------------------------------------------
#ifndef LN_BWD_ROWS_DEF
#define LN_BWD_ROWS_DEF 16
#endif
constexpr int LN_BWD_ROWS = LN_BWD_ROWS_DEF;      // rows per block (4 waves x 4 rows): twice the blocks of the first version -- a wave's rows are a serial
                                      // chain of load -> reduce -> store, more waves in flight hide more of it

__global__ void __launch_bounds__(256) add_ln_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ d_out_pos,
                                                          const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                          const float* __restrict__ rstd_i, float* __restrict__ d_x,
                                                          float* __restrict__ d_y, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int rows, int C, uint32_t thresh,
                                                          float inv_keep, const uint64_t* rng_state, uint32_t rng_stream) {
  __shared__ float red[2][4][1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = C >> 8, rem = (C & 255) >> 2;
  uint64_t seed = 0, step = 0;
  if (thresh) { seed = rng_state[0]; step = rng_state[1]; }
  float4 ag[LN_MAXV], ab[LN_MAXV], gam[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool act = (i < nv) || (i == nv && lane < rem);
    gam[i] = act ? *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r_begin = blockIdx.x * LN_BWD_ROWS;
  for (int rr = w; rr < LN_BWD_ROWS; rr += 4) {
    const int row = r_begin + rr;
    if (row >= rows) break;
    const long long base = (long long)row * C;
    const float mean = mean_i[row], rstd = rstd_i[row];
    float4 xh[LN_MAXV], g[LN_MAXV];
    uint32_t keepbits = 0;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const bool act = (i < nv) || (i == nv && lane < rem);
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (act) {
        const int c = (i * 64 + lane) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + base + c);
        if (y) {
          float4 q = *reinterpret_cast<const float4*>(y + base + c);
          if (thresh) {
            const uint64_t idx = (uint64_t)base + c;
            const bool k0 = cape_keep(seed, step, rng_stream, idx + 0, thresh);
            const bool k1 = cape_keep(seed, step, rng_stream, idx + 1, thresh);
            const bool k2 = cape_keep(seed, step, rng_stream, idx + 2, thresh);
            const bool k3 = cape_keep(seed, step, rng_stream, idx + 3, thresh);
            keepbits |= ((uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2) | ((uint32_t)k3 << 3)) << (4 * i);
            q.x = k0 ? q.x * inv_keep : 0.f; q.y = k1 ? q.y * inv_keep : 0.f;
            q.z = k2 ? q.z * inv_keep : 0.f; q.w = k3 ? q.w * inv_keep : 0.f;
          }
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        float4 d = *reinterpret_cast<const float4*>(d_out + base + c);
        if (d_out_pos) {
          const float4 e = *reinterpret_cast<const float4*>(d_out_pos + base + c);
          d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
        }
        xh[i].x = (v.x - mean) * rstd; xh[i].y = (v.y - mean) * rstd;
        xh[i].z = (v.z - mean) * rstd; xh[i].w = (v.w - mean) * rstd;
        ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        g[i].x = d.x * gam[i].x; g[i].y = d.y * gam[i].y; g[i].z = d.z * gam[i].z; g[i].w = d.w * gam[i].w;
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const bool act = (i < nv) || (i == nv && lane < rem);
      if (act) {
        const int c = (i * 64 + lane) * 4;
        float4 ds;
        ds.x = rstd * (g[i].x - m1 - xh[i].x * m2);
        ds.y = rstd * (g[i].y - m1 - xh[i].y * m2);
        ds.z = rstd * (g[i].z - m1 - xh[i].z * m2);
        ds.w = rstd * (g[i].w - m1 - xh[i].w * m2);
        *reinterpret_cast<float4*>(d_x + base + c) = ds;
        if (d_y && d_y != d_x) {
          if (thresh) {
            const uint32_t kb = keepbits >> (4 * i);
            ds.x = (kb & 1) ? ds.x * inv_keep : 0.f; ds.y = (kb & 2) ? ds.y * inv_keep : 0.f;
            ds.z = (kb & 4) ? ds.z * inv_keep : 0.f; ds.w = (kb & 8) ? ds.w * inv_keep : 0.f;
          }
          *reinterpret_cast<float4*>(d_y + base + c) = ds;
        }
      }
    }
  }
  // combine the 4 waves' dgamma/dbeta partials
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool act = (i < nv) || (i == nv && lane < rem);
    if (act) {
      const int c = (i * 64 + lane) * 4;
      *reinterpret_cast<float4*>(&red[0][w][c]) = ag[i];
      *reinterpret_cast<float4*>(&red[1][w][c]) = ab[i];
    }
  }
  __syncthreads();
#ifndef LAB_LN_NO_ATOMIC
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(&dgamma[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(&dbeta[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The model width (C == 256): one float4 per lane and tensor, so a wave takes FOUR rows per trip with all of their loads in
// flight together (the general kernel above walks its rows one by one: load -> two wave reductions -> store, a dependent
// chain per row that left it at a third of the HBM rate: 75 us for 43520 rows against 22 us for the forward).  Blocks stride
// over the rows, keep their dgamma / dbeta partial sums in registers for the whole launch and add them once (the per-16-row
// atomics of the general kernel put 87 k adds on each of the 16 hot cache lines: 28 us of kernel became 41-76 us, depending on
// the rows per block; here 256 blocks add once).
// ---------------------------------------------------------------------------------------------
constexpr int LN256_R = 4;             // rows per wave and trip

constexpr int LN256_W = 16;            // waves per block: ONE block per CU, so that only 256 blocks add their dgamma / dbeta partials

__global__ void __launch_bounds__(64 * LN256_W) add_ln_bwd256_kernel(const float* __restrict__ d_out, const float* __restrict__ d_out_pos,
                                                             const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                             const float* __restrict__ rstd_i, float* __restrict__ d_x,
                                                             float* __restrict__ d_y, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int rows, uint32_t thresh, float inv_keep,
                                                             const uint64_t* rng_state, uint32_t rng_stream) {
  __shared__ float red[2][LN256_W][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t seed = 0, step = 0;
  if (thresh) { seed = rng_state[0]; step = rng_state[1]; }
  const float4 gam = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool split = d_y && d_y != d_x;
  const int wave_g = blockIdx.x * LN256_W + w, nwaves = gridDim.x * LN256_W;
  for (int r0 = wave_g * LN256_R; r0 < rows; r0 += nwaves * LN256_R) {
    float4 v[LN256_R], d[LN256_R];
    float mean[LN256_R], rstd[LN256_R];
    uint32_t kb[LN256_R];
#pragma unroll
    for (int k = 0; k < LN256_R; ++k) {                          // every load of the trip is requested before the first use
      const int row = min(r0 + k, rows - 1);
      const long long base = (long long)row * 256 + 4 * lane;
      v[k] = *reinterpret_cast<const float4*>(x + base);
      d[k] = *reinterpret_cast<const float4*>(d_out + base);
      mean[k] = mean_i[row]; rstd[k] = rstd_i[row];
    }
    if (y) {
#pragma unroll
      for (int k = 0; k < LN256_R; ++k) {
        const int row = min(r0 + k, rows - 1);
        const long long base = (long long)row * 256 + 4 * lane;
        float4 q = *reinterpret_cast<const float4*>(y + base);
        kb[k] = 0xF;
        if (thresh) {
          const uint64_t idx = (uint64_t)base;
          const bool k0 = cape_keep(seed, step, rng_stream, idx + 0, thresh), k1 = cape_keep(seed, step, rng_stream, idx + 1, thresh);
          const bool k2 = cape_keep(seed, step, rng_stream, idx + 2, thresh), k3 = cape_keep(seed, step, rng_stream, idx + 3, thresh);
          kb[k] = (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2) | ((uint32_t)k3 << 3);
          q.x = k0 ? q.x * inv_keep : 0.f; q.y = k1 ? q.y * inv_keep : 0.f;
          q.z = k2 ? q.z * inv_keep : 0.f; q.w = k3 ? q.w * inv_keep : 0.f;
        }
        v[k].x += q.x; v[k].y += q.y; v[k].z += q.z; v[k].w += q.w;
      }
    }
    if (d_out_pos) {
#pragma unroll
      for (int k = 0; k < LN256_R; ++k) {
        const int row = min(r0 + k, rows - 1);
        const float4 e = *reinterpret_cast<const float4*>(d_out_pos + (long long)row * 256 + 4 * lane);
        d[k].x += e.x; d[k].y += e.y; d[k].z += e.z; d[k].w += e.w;
      }
    }
#pragma unroll
    for (int k = 0; k < LN256_R; ++k) {
      const bool live = r0 + k < rows;                            // wave-uniform
      float4 xh, g;
      xh.x = (v[k].x - mean[k]) * rstd[k]; xh.y = (v[k].y - mean[k]) * rstd[k];
      xh.z = (v[k].z - mean[k]) * rstd[k]; xh.w = (v[k].w - mean[k]) * rstd[k];
      if (live) {
        ag.x += d[k].x * xh.x; ag.y += d[k].y * xh.y; ag.z += d[k].z * xh.z; ag.w += d[k].w * xh.w;
        ab.x += d[k].x; ab.y += d[k].y; ab.z += d[k].z; ab.w += d[k].w;
      }
      g.x = d[k].x * gam.x; g.y = d[k].y * gam.y; g.z = d[k].z * gam.z; g.w = d[k].w * gam.w;
      const float m1 = wave_sum(g.x + g.y + g.z + g.w) * (1.f / 256.f);
      const float m2 = wave_sum(g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w) * (1.f / 256.f);
      float4 ds;
      ds.x = rstd[k] * (g.x - m1 - xh.x * m2); ds.y = rstd[k] * (g.y - m1 - xh.y * m2);
      ds.z = rstd[k] * (g.z - m1 - xh.z * m2); ds.w = rstd[k] * (g.w - m1 - xh.w * m2);
      if (live) {
        const long long base = (long long)(r0 + k) * 256 + 4 * lane;
        *reinterpret_cast<float4*>(d_x + base) = ds;
        if (split) {
          if (thresh) {
            ds.x = (kb[k] & 1) ? ds.x * inv_keep : 0.f; ds.y = (kb[k] & 2) ? ds.y * inv_keep : 0.f;
            ds.z = (kb[k] & 4) ? ds.z * inv_keep : 0.f; ds.w = (kb[k] & 8) ? ds.w * inv_keep : 0.f;
          }
          *reinterpret_cast<float4*>(d_y + base) = ds;
        }
      }
    }
  }
  *reinterpret_cast<float4*>(&red[0][w][4 * lane]) = ag;
  *reinterpret_cast<float4*>(&red[1][w][4 * lane]) = ab;
  __syncthreads();
  if (threadIdx.x < 512) {
    const int which = threadIdx.x >> 8, c = threadIdx.x & 255;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < LN256_W; ++k) a += red[which][k][c];
    atomicAdd((which ? dbeta : dgamma) + c, a);
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm over NHWC, two launches each way so that a level is spread over (N x row chunks) blocks instead of N
// (the one-block-per-image form ran 32 blocks on 256 CUs: 240 us per level-0 call).  Thread = channel (full 1 KB rows,
// coalesced), GN_ROWS rows per block.
//   forward : stats kernel -- per-thread fp32 partial sum / sum of squares over <= GN_ROWS rows, reduced over the
//             group's lanes, then ONE fp64 atomic pair per (block, group) into ws[n][g][2]; apply kernel -- mean and
//             variance from the fp64 sums (E[x^2] - E[x]^2 evaluated in double), normalise the chunk.
//   backward: stats kernel -- per-channel sum(dy), sum(dy * xhat) into ws[n][c][2] (fp32 atomics, <= HW/GN_ROWS adds per
//             address); apply kernel -- group terms from those sums; the chunk-0 block of every image adds the image's
//             totals to dgamma / dbeta (N adds per address).
// ---------------------------------------------------------------------------------------------
constexpr int GN_ROWS = 16;

__device__ __forceinline__ float group_sum(float v, int cpg) {
  for (int o = 1; o < cpg; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void groupnorm_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C, int G) {
  const int n = blockIdx.y, c = threadIdx.x;
  const int cpg = C / G;
  const int r0 = blockIdx.x * GN_ROWS, r1 = min(HW, r0 + GN_ROWS);
  const float* xp = x + (long long)n * HW * C + c;
  float s = 0.f, q = 0.f;
  for (int r = r0; r < r1; ++r) { const float v = xp[(long long)r * C]; s += v; q += v * v; }
  s = group_sum(s, cpg); q = group_sum(q, cpg);
  if ((c % cpg) == 0) {
    double* w = ws + ((long long)n * G + c / cpg) * 2;
    atomicAdd(w, (double)s);
    atomicAdd(w + 1, (double)q);
  }
}

__global__ void groupnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ out, long long out_image_stride,
                                       const double* __restrict__ ws, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                       int HW, int C, int G) {
  const int n = blockIdx.y, c = threadIdx.x;
  const int cpg = C / G, g = c / cpg;
  const double cnt = (double)HW * (double)cpg;
  const double m = ws[((long long)n * G + g) * 2] / cnt;
  const double var = fmax(ws[((long long)n * G + g) * 2 + 1] / cnt - m * m, 0.0);
  const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + 1e-5));
  if (blockIdx.x == 0 && (c % cpg) == 0) { mean_o[n * G + g] = mean; rstd_o[n * G + g] = rstd; }
  const float ga = gamma[c] * rstd, be = beta[c] - mean * gamma[c] * rstd;
  const int r0 = blockIdx.x * GN_ROWS, r1 = min(HW, r0 + GN_ROWS);
  const float* xp = x + (long long)n * HW * C + c;
  float* op = out + (long long)n * out_image_stride + c;
  for (int r = r0; r < r1; ++r) op[(long long)r * C] = xp[(long long)r * C] * ga + be;
}

__global__ void groupnorm_bwd_stats_kernel(const float* __restrict__ d_out, long long d_out_image_stride,
                                           const float* __restrict__ x, const float* __restrict__ mean_i,
                                           const float* __restrict__ rstd_i, float* __restrict__ ws, int HW, int C, int G) {
  const int n = blockIdx.y, c = threadIdx.x;
  const int cpg = C / G;
  const float mean = mean_i[n * G + c / cpg], rstd = rstd_i[n * G + c / cpg];
  const int r0 = blockIdx.x * GN_ROWS, r1 = min(HW, r0 + GN_ROWS);
  const float* xp = x + (long long)n * HW * C + c;
  const float* dp = d_out + (long long)n * d_out_image_stride + c;
  float sd = 0.f, sdx = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float d = dp[(long long)r * C];
    sd += d;
    sdx += d * (xp[(long long)r * C] - mean) * rstd;
  }
  float* w = ws + ((long long)n * C + c) * 2;
  atomicAdd(w, sd);
  atomicAdd(w + 1, sdx);
}

__global__ void groupnorm_bwd_apply_kernel(const float* __restrict__ d_out, long long d_out_image_stride,
                                           const float* __restrict__ x, const float* __restrict__ gamma,
                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                           const float* __restrict__ ws, float* __restrict__ d_x,
                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int HW, int C, int G) {
  const int n = blockIdx.y, c = threadIdx.x;
  const int cpg = C / G;
  const float mean = mean_i[n * G + c / cpg], rstd = rstd_i[n * G + c / cpg];
  const float sd = ws[((long long)n * C + c) * 2], sdx = ws[((long long)n * C + c) * 2 + 1];
  if (blockIdx.x == 0) { atomicAdd(&dgamma[c], sdx); atomicAdd(&dbeta[c], sd); }
  const float g = gamma[c];
  const float cnt = (float)HW * (float)cpg;
  const float A = group_sum(g * sd, cpg) / cnt;
  const float B = group_sum(g * sdx, cpg) / cnt;
  const int r0 = blockIdx.x * GN_ROWS, r1 = min(HW, r0 + GN_ROWS);
  const float* xp = x + (long long)n * HW * C + c;
  const float* dp = d_out + (long long)n * d_out_image_stride + c;
  float* op = d_x + (long long)n * HW * C + c;
  for (int r = r0; r < r1; ++r) {
    const float xh = (xp[(long long)r * C] - mean) * rstd;
    op[(long long)r * C] = rstd * (g * dp[(long long)r * C] - A - xh * B);
  }
}

}  // namespace

extern "C" int cape_add_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta,
                                      float* out, float* mean, float* rstd, const float* pos, float* out_pos,
                                      int rows, int C, float dropout_p, const uint64_t* rng_state,
                                      uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(x && gamma && beta && out && mean && rstd, "cape_add_layernorm_fwd: null pointer");
  CAPE_REQUIRE(C > 0 && C <= 1024 && (C % 4) == 0, "cape_add_layernorm_fwd: C=%d must be a multiple of 4, <= 1024", C);
  CAPE_REQUIRE((pos == nullptr) == (out_pos == nullptr), "cape_add_layernorm_fwd: pos and out_pos go together");
  CAPE_REQUIRE(dropout_p == 0.f || (y && rng_state && dropout_p < 1.f), "cape_add_layernorm_fwd: dropout needs y and rng_state");
  if (rows <= 0) return 0;
  const uint32_t th = dropout_p > 0.f ? cape_drop_threshold(dropout_p) : 0u;
  const float ik = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  hipLaunchKernelGGL(add_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, y, gamma, beta, out,
                     mean, rstd, pos, out_pos, rows, C, th, ik, rng_state, rng_stream);
  CAPE_LAUNCH_CHECK("cape_add_layernorm_fwd");
  return 0;
}

extern "C" int cape_add_layernorm_bwd(const float* d_out, const float* d_out_pos, const float* x, const float* y,
                                      const float* gamma, const float* mean, const float* rstd, float* d_x,
                                      float* d_y, float* dgamma, float* dbeta, int rows, int C, float dropout_p,
                                      const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(d_out && x && gamma && mean && rstd && d_x && dgamma && dbeta, "cape_add_layernorm_bwd: null pointer");
  CAPE_REQUIRE(C > 0 && C <= 1024 && (C % 4) == 0, "cape_add_layernorm_bwd: C=%d must be a multiple of 4, <= 1024", C);
  CAPE_REQUIRE(dropout_p == 0.f || (y && d_y && d_y != d_x && rng_state && dropout_p < 1.f),
               "cape_add_layernorm_bwd: dropout needs y, a separate d_y and rng_state");
  if (rows <= 0) return 0;
  const uint32_t th = dropout_p > 0.f ? cape_drop_threshold(dropout_p) : 0u;
  const float ik = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  static const bool general_only = getenv("CAPE_LN_BWD_GENERAL") != nullptr;      // tuning switch
  if (C == 256 && !general_only) {
    const int trips = (rows + LN256_W * LN256_R - 1) / (LN256_W * LN256_R);      // row groups of one block-trip
    const int blocks = trips < 256 ? trips : 256;                    // one 16-wave block per CU; more rows -> more trips per block
    hipLaunchKernelGGL(add_ln_bwd256_kernel, dim3(blocks), dim3(64 * LN256_W), 0, as_stream(stream), d_out, d_out_pos, x, y, gamma, mean, rstd,
                       d_x, d_y, dgamma, dbeta, rows, th, ik, rng_state, rng_stream);
    CAPE_LAUNCH_CHECK("cape_add_layernorm_bwd");
    return 0;
  }
  hipLaunchKernelGGL(add_ln_bwd_kernel, dim3((rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS), dim3(256), 0, as_stream(stream),
                     d_out, d_out_pos, x, y, gamma, mean, rstd, d_x, d_y, dgamma, dbeta, rows, C, th, ik, rng_state,
                     rng_stream);
  CAPE_LAUNCH_CHECK("cape_add_layernorm_bwd");
  return 0;
}

static int gn_check(int N, int HW, int C, int G) {
  if (N <= 0 || HW <= 0) return 0;
  CAPE_REQUIRE(C >= 64 && C <= 1024 && (C % 64) == 0, "cape_groupnorm: C=%d must be a multiple of 64, <= 1024", C);
  CAPE_REQUIRE(G > 0 && (C % G) == 0, "cape_groupnorm: G=%d must divide C=%d", G, C);
  const int cpg = C / G;
  CAPE_REQUIRE(cpg <= 64 && (cpg & (cpg - 1)) == 0, "cape_groupnorm: channels per group (%d) must be a power of two <= 64", cpg);
  return 0;
}

extern "C" size_t cape_groupnorm_workspace_bytes(int N, int C, int G) {
  if (N <= 0 || C <= 0 || G <= 0) return 0;
  const size_t f = (size_t)N * G * 2 * sizeof(double), b = (size_t)N * C * 2 * sizeof(float);
  return f > b ? f : b;
}

extern "C" int cape_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* out,
                                  long long out_image_stride, float* mean, float* rstd, int N, int HW, int C, int G,
                                  void* workspace, size_t workspace_bytes, cape_stream_t stream) {
  CAPE_REQUIRE(x && gamma && beta && out && mean && rstd, "cape_groupnorm_fwd: null pointer");
  if (gn_check(N, HW, C, G)) return 1;
  if (N <= 0 || HW <= 0) return 0;
  const size_t need = (size_t)N * G * 2 * sizeof(double);
  CAPE_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
               "cape_groupnorm_fwd: workspace of %zu bytes (8-byte aligned) needed, %zu given", need, workspace_bytes);
  CAPE_REQUIRE(N <= 65535, "cape_groupnorm_fwd: N too large");
  hipError_t e = hipMemsetAsync(workspace, 0, need, as_stream(stream));
  if (e != hipSuccess) return cape_set_error("cape_groupnorm_fwd: memset: %s", hipGetErrorString(e));
  const dim3 grid((HW + GN_ROWS - 1) / GN_ROWS, N);
  hipLaunchKernelGGL(groupnorm_stats_kernel, grid, dim3(C), 0, as_stream(stream), x, static_cast<double*>(workspace), HW, C, G);
  hipLaunchKernelGGL(groupnorm_apply_kernel, grid, dim3(C), 0, as_stream(stream), x, gamma, beta, out, out_image_stride,
                     static_cast<const double*>(workspace), mean, rstd, HW, C, G);
  CAPE_LAUNCH_CHECK("cape_groupnorm_fwd");
  return 0;
}

extern "C" int cape_groupnorm_bwd(const float* d_out, long long d_out_image_stride, const float* x,
                                  const float* gamma, const float* mean, const float* rstd, float* d_x,
                                  float* dgamma, float* dbeta, int N, int HW, int C, int G, void* workspace,
                                  size_t workspace_bytes, cape_stream_t stream) {
  CAPE_REQUIRE(d_out && x && gamma && mean && rstd && d_x && dgamma && dbeta, "cape_groupnorm_bwd: null pointer");
  if (gn_check(N, HW, C, G)) return 1;
  if (N <= 0 || HW <= 0) return 0;
  const size_t need = (size_t)N * C * 2 * sizeof(float);
  CAPE_REQUIRE(workspace && workspace_bytes >= need, "cape_groupnorm_bwd: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
  CAPE_REQUIRE(N <= 65535, "cape_groupnorm_bwd: N too large");
  hipError_t e = hipMemsetAsync(workspace, 0, need, as_stream(stream));
  if (e != hipSuccess) return cape_set_error("cape_groupnorm_bwd: memset: %s", hipGetErrorString(e));
  const dim3 grid((HW + GN_ROWS - 1) / GN_ROWS, N);
  hipLaunchKernelGGL(groupnorm_bwd_stats_kernel, grid, dim3(C), 0, as_stream(stream), d_out, d_out_image_stride, x, mean, rstd,
                     static_cast<float*>(workspace), HW, C, G);
  hipLaunchKernelGGL(groupnorm_bwd_apply_kernel, grid, dim3(C), 0, as_stream(stream), d_out, d_out_image_stride, x, gamma, mean,
                     rstd, static_cast<const float*>(workspace), d_x, dgamma, dbeta, HW, C, G);
  CAPE_LAUNCH_CHECK("cape_groupnorm_bwd");
  return 0;
}
