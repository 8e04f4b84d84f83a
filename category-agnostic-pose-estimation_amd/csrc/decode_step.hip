// decode_step.hip -- the cached autoregressive decode step as ~12 launches per decoder layer instead of ~31
// (RoomFormerV2.forward_inference -> TransformerDecoder / TransformerDecoderLayer v1 with one query token per image,
// reference models/roomformer_v2.py:481-598, deformable_transformer_v2.py:320-370, :1024-1131).
//
// A step multiplies N <= 64 token rows (N = images in flight) by every weight of the decoder: 34 MB of fp32 weights against
// a few KB of activations -- weight streaming with a dependency between every two products, i.e. latency bound.  Two
// kernels carry the step (the single-query attention and the one-query MSDA gather keep their own kernels):
//
//   cape_decode_linear   out = [relu] ( LNin(X) [+ add] ) W^T [+ X2 W2^T on the first n2 columns] + b [+ LNres(R)]
//       * column-split: a block owns 8 output columns and all N rows, so every weight is read exactly once per step;
//       * LayerNorm "on load": the post-norm layer structure x_{k+1} = LN(x_k + f(x_k)) is kept as *pre-norm sums* in
//         memory; each consumer normalises the N x 256 rows itself while it stages them into LDS (a few thousand flops,
//         redundant across blocks) -- the four LayerNorm launches per layer disappear and so do their round trips;
//       * the residual operand is normalised the same way;
//       * a second product on the first n2 columns carries the `+ query_pos` of the self-attention query through the
//         folded projection (q = (attn_q(t) + pos) Wq^T = t (Wq Wa)^T + pos Wq^T), so q, k, v of a layer are ONE launch
//         whose k / v columns land directly in row `step` of the KV cache (three output segments with their own strides).
//   cape_decode_tail     everything between two layers, one block per image row: LN3 -> coords MLP (256-256-256-2) ->
//       reference refinement sigmoid(delta + logit(ref)) -> [class head on the last layer] -> next layer's query position
//       embedding LN(pos_trans(sine(ref'))) and level-scaled reference points.  Row-split on purpose: the five products
//       depend on each other, 0.8 MB of weights per block from L2 costs less than five launch boundaries.
//
// All arithmetic is plain fp32 FMA (exact fp32, like the skinny kernel of gemm.hip): the decode path does not use the
// bf16x3 split.
#include "common.h"

namespace {

constexpr int DL_COLS = 8;          // output columns per block
constexpr int DL_MAXN = 64;         // rows
constexpr int DL_KC = 256;          // k chunk staged in LDS

struct DecLinP {
  int N, K, Nout;
  const float* X; long long ldx; const float* in_gamma; const float* in_beta; const float* in_add; long long ld_add;
  const float* W; long long ldw; const float* bias;
  const float* X2; long long ldx2; int K2; const float* W2; long long ldw2; int n2;
  const float* R; long long ldr; const float* res_gamma; const float* res_beta;
  int relu;
  int nseg, seg; float* out[3]; long long ldo[3];
};

// mean / rstd of N rows of width C (C <= 1024, C % 4 == 0): wave w takes rows w, w+8, ...
__device__ __forceinline__ void row_stats(const float* src, long long ld, int N, int C, float* mean_s, float* rstd_s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < N; r += 8) {
    const float* p = src + (long long)r * ld;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(p + c);
      s += v.x + v.y + v.z + v.w;
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(p + c);
      const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
      q += a * a + b * b + d * d + e * e;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
    if (lane == 0) { mean_s[r] = mean; rstd_s[r] = rstd; }
  }
}

__global__ void __launch_bounds__(512) decode_linear_kernel(const DecLinP p) {
  extern __shared__ __attribute__((aligned(16))) float dl_lds[];              // xs[N][260] | ws[8][260] | 4 x stats[N]
  float* xs = dl_lds;
  float* ws = xs + p.N * (DL_KC + 4);
  float* mean_i = ws + DL_COLS * (DL_KC + 4);
  float* rstd_i = mean_i + p.N;
  float* mean_r = rstd_i + p.N;
  float* rstd_r = mean_r + p.N;
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * DL_COLS;
  const int col = t & (DL_COLS - 1), row = t >> 3;
  constexpr int LD = DL_KC + 4;
  if (p.in_gamma) row_stats(p.X, p.ldx, p.N, p.K, mean_i, rstd_i);
  if (p.res_gamma) row_stats(p.R, p.ldr, p.N, p.Nout, mean_r, rstd_r);
  if (p.in_gamma || p.res_gamma) __syncthreads();
  float acc = 0.f;
  // products: pass 0 = X W^T over K, pass 1 = X2 W2^T over K2 (only for blocks whose columns lie below n2)
  const int npass = (p.X2 && n0 < p.n2) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const float* X = pass ? p.X2 : p.X;
    const long long ldx = pass ? p.ldx2 : p.ldx;
    const float* W = pass ? p.W2 : p.W;
    const long long ldw = pass ? p.ldw2 : p.ldw;
    const int K = pass ? p.K2 : p.K;
    const bool ln = !pass && p.in_gamma;
    for (int k0 = 0; k0 < K; k0 += DL_KC) {
      const int kc = min(DL_KC, K - k0), kq = kc >> 2;
      if (pass || k0) __syncthreads();
      for (int i = t; i < p.N * kq; i += 512) {
        const int r = i / kq, c = (i - r * kq) * 4;
        float4 v = *reinterpret_cast<const float4*>(X + (long long)r * ldx + k0 + c);
        if (ln) {
          const float4 g = *reinterpret_cast<const float4*>(p.in_gamma + k0 + c);
          const float4 b = *reinterpret_cast<const float4*>(p.in_beta + k0 + c);
          const float m = mean_i[r], s = rstd_i[r];
          v.x = (v.x - m) * s * g.x + b.x; v.y = (v.y - m) * s * g.y + b.y;
          v.z = (v.z - m) * s * g.z + b.z; v.w = (v.w - m) * s * g.w + b.w;
        }
        if (!pass && p.in_add) {
          const float4 a = *reinterpret_cast<const float4*>(p.in_add + (long long)r * p.ld_add + k0 + c);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(&xs[r * LD + c]) = v;
      }
      for (int i = t; i < DL_COLS * kq; i += 512) {
        const int r = i / kq, c = (i - r * kq) * 4;
        const int n = min(n0 + r, p.Nout - 1);
        *reinterpret_cast<float4*>(&ws[r * LD + c]) = *reinterpret_cast<const float4*>(W + (long long)n * ldw + k0 + c);
      }
      __syncthreads();
      if (row < p.N) {
        const float* xr = &xs[row * LD];
        const float* wr = &ws[col * LD];
#pragma unroll 4
        for (int k = 0; k < kc; k += 4) {
          const float4 a = *reinterpret_cast<const float4*>(xr + k);
          const float4 b = *reinterpret_cast<const float4*>(wr + k);
          acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
      }
    }
  }
  const int n = n0 + col;
  if (row < p.N && n < p.Nout) {
    float v = acc + (p.bias ? p.bias[n] : 0.f);
    if (p.R) {
      float r = p.R[(long long)row * p.ldr + n];
      if (p.res_gamma) r = (r - mean_r[row]) * rstd_r[row] * p.res_gamma[n] + p.res_beta[n];
      v += r;
    }
    if (p.relu) v = fmaxf(v, 0.f);
    const int sg = n / p.seg;
    p.out[sg][(long long)row * p.ldo[sg] + (n - sg * p.seg)] = v;
  }
}

// ------------------------------------------------------------------------------------------------
struct DecTailP {
  int N, L, last;
  const float* P4; long long ldp; const float* g3; const float* b3;       // pre-norm output of the layer + its norm3
  const float* W1; const float* B1; const float* W2; const float* B2; const float* W3; const float* B3;   // coords MLP
  const float* ref;                                                        // (N, 2) reference points of this layer
  const float* Wc; const float* Bc; int ncls;                              // class head (last layer) or null
  const float* Wp; const float* Bp; const float* gp; const float* bp;      // pos_trans + pos_trans_norm (next layer) or null
  const float* dim_t;                                                      // 128 sine periods
  const float* vr;                                                         // (N, L, 2) valid ratios
  float* ref_out; long long ld_ref;                                        // refined points -> (N, 2) / a slot of out_coords
  float* qpos_out;                                                         // (N, 256) next layer's query position embedding
  float* refin_out;                                                        // (N, L, 2) next layer's level-scaled points
  float* cls_out; long long ld_cls;                                        // class logits -> a slot of out_logits
  float* hs_out; long long ld_hs;                                          // LN3 output (last layer, for room logits) or null
};

__device__ __forceinline__ float inv_sigmoid_f(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  return logf(x1 / x2);
}

// y[j] = act(b[j] + W[j][:] . x) for j = wave, wave + 8, ... < nout; x lives in registers (4 consecutive k per lane)
__device__ __forceinline__ void wave_gemv(const float* W, const float* B, const float4 x, float* y, int nout, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < nout; j += 8) {
    const float4 w = *reinterpret_cast<const float4*>(W + (long long)j * 256 + 4 * lane);
    float s = fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, w.w * x.w)));
    s = wave_sum(s);
    if (lane == 0) { s += B ? B[j] : 0.f; y[j] = relu ? fmaxf(s, 0.f) : s; }
  }
}

__global__ void __launch_bounds__(512) decode_tail_kernel(const DecTailP p) {
  __shared__ __attribute__((aligned(16))) float buf[2][256];
  __shared__ float small[8];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // LN3 of this row, redundantly in every wave (no barrier): x = 4 consecutive channels per lane
  float4 x = *reinterpret_cast<const float4*>(p.P4 + (long long)n * p.ldp + 4 * lane);
  {
    const float mean = wave_sum(x.x + x.y + x.z + x.w) / 256.f;
    const float a = x.x - mean, b = x.y - mean, c = x.z - mean, d = x.w - mean;
    const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) / 256.f + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(p.g3 + 4 * lane);
    const float4 be = *reinterpret_cast<const float4*>(p.b3 + 4 * lane);
    x = make_float4(a * rstd * g.x + be.x, b * rstd * g.y + be.y, c * rstd * g.z + be.z, d * rstd * g.w + be.w);
  }
  if (p.hs_out && wave == 0) *reinterpret_cast<float4*>(p.hs_out + (long long)n * p.ld_hs + 4 * lane) = x;
  if (p.Wc) wave_gemv(p.Wc, p.Bc, x, small + 2, p.ncls, false);            // class head reads the layer output
  wave_gemv(p.W1, p.B1, x, buf[0], 256, true);
  __syncthreads();
  float4 h = *reinterpret_cast<const float4*>(&buf[0][4 * lane]);
  wave_gemv(p.W2, p.B2, h, buf[1], 256, true);
  __syncthreads();
  h = *reinterpret_cast<const float4*>(&buf[1][4 * lane]);
  wave_gemv(p.W3, p.B3, h, small, 2, false);
  __syncthreads();
  if (threadIdx.x < 2) {
    const float z = small[threadIdx.x] + inv_sigmoid_f(p.ref[n * 2 + threadIdx.x]);
    const float r = 1.f / (1.f + expf(-z));
    small[threadIdx.x] = r;
    p.ref_out[(long long)n * p.ld_ref + threadIdx.x] = r;
  }
  if (p.cls_out && threadIdx.x < p.ncls) p.cls_out[(long long)n * p.ld_cls + threadIdx.x] = small[2 + threadIdx.x];
  if (!p.Wp) return;
  __syncthreads();
  const float rx = small[0], ry = small[1];
  if (threadIdx.x < 2 * p.L) {                                               // next layer's reference points per level
    const int l = threadIdx.x >> 1, a = threadIdx.x & 1;
    p.refin_out[((long long)n * p.L + l) * 2 + a] = (a ? ry : rx) * p.vr[((long long)n * p.L + l) * 2 + a];
  }
  // sine embedding of the refined point (channel c: axis c >> 7, period dim_t[c & 127], odd -> cos), 4 channels per lane
  float e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * lane + i, k = c & 127;
    const float v = ((c >> 7) ? ry : rx) * 6.283185307179586f / p.dim_t[k];
    e[i] = (k & 1) ? cosf(v) : sinf(v);
  }
  wave_gemv(p.Wp, p.Bp, make_float4(e[0], e[1], e[2], e[3]), buf[0], 256, false);
  __syncthreads();
  if (wave == 0) {
    const float4 q = *reinterpret_cast<const float4*>(&buf[0][4 * lane]);
    const float mean = wave_sum(q.x + q.y + q.z + q.w) / 256.f;
    const float a = q.x - mean, b = q.y - mean, c = q.z - mean, d = q.w - mean;
    const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) / 256.f + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(p.gp + 4 * lane);
    const float4 be = *reinterpret_cast<const float4*>(p.bp + 4 * lane);
    *reinterpret_cast<float4*>(p.qpos_out + (long long)n * 256 + 4 * lane) =
        make_float4(a * rstd * g.x + be.x, b * rstd * g.y + be.y, c * rstd * g.z + be.z, d * rstd * g.w + be.w);
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int cape_decode_linear(const cape_decode_linear_desc* d, cape_stream_t stream) {
  CAPE_REQUIRE(d != nullptr, "cape_decode_linear: null descriptor");
  CAPE_REQUIRE(d->N >= 1 && d->N <= DL_MAXN, "cape_decode_linear: N=%d rows, at most %d", d->N, DL_MAXN);
  CAPE_REQUIRE(d->K > 0 && d->K % 4 == 0 && d->Nout > 0, "cape_decode_linear: K=%d must be a positive multiple of 4", d->K);
  CAPE_REQUIRE(d->X && d->W && al16(d->X) && al16(d->W) && d->ldx % 4 == 0 && d->ldw % 4 == 0, "cape_decode_linear: X / W must be 16-byte aligned rows");
  CAPE_REQUIRE((d->in_gamma != nullptr) == (d->in_beta != nullptr) && (d->res_gamma != nullptr) == (d->res_beta != nullptr),
               "cape_decode_linear: LayerNorm parameters come in (gamma, beta) pairs");
  if (d->in_gamma) CAPE_REQUIRE(d->K <= 1024 && al16(d->in_gamma) && al16(d->in_beta), "cape_decode_linear: LN-on-load needs K <= 1024");
  if (d->in_add) CAPE_REQUIRE(al16(d->in_add) && d->ld_add % 4 == 0, "cape_decode_linear: in_add must be 16-byte aligned rows");
  if (d->res_gamma) CAPE_REQUIRE(d->R && d->Nout % 4 == 0 && d->Nout <= 1024 && al16(d->R) && d->ldr % 4 == 0, "cape_decode_linear: normalised residual needs aligned rows of width Nout <= 1024");
  if (d->X2) CAPE_REQUIRE(d->W2 && d->K2 > 0 && d->K2 % 4 == 0 && d->n2 > 0 && d->n2 % DL_COLS == 0 && d->n2 <= d->Nout && al16(d->X2) && al16(d->W2) &&
                          d->ldx2 % 4 == 0 && d->ldw2 % 4 == 0, "cape_decode_linear: bad second product");
  CAPE_REQUIRE(d->nseg >= 1 && d->nseg <= 3 && d->seg > 0 && d->nseg * d->seg == d->Nout, "cape_decode_linear: output segments must tile Nout");
  for (int i = 0; i < d->nseg; ++i) CAPE_REQUIRE(d->out[i] != nullptr, "cape_decode_linear: null output segment");
  DecLinP p;
  p.N = d->N; p.K = d->K; p.Nout = d->Nout;
  p.X = d->X; p.ldx = d->ldx; p.in_gamma = d->in_gamma; p.in_beta = d->in_beta; p.in_add = d->in_add; p.ld_add = d->ld_add;
  p.W = d->W; p.ldw = d->ldw; p.bias = d->bias;
  p.X2 = d->X2; p.ldx2 = d->ldx2; p.K2 = d->K2; p.W2 = d->W2; p.ldw2 = d->ldw2; p.n2 = d->X2 ? d->n2 : 0;
  p.R = d->R; p.ldr = d->ldr; p.res_gamma = d->res_gamma; p.res_beta = d->res_beta;
  p.relu = d->relu; p.nseg = d->nseg; p.seg = d->seg;
  for (int i = 0; i < 3; ++i) { p.out[i] = i < d->nseg ? d->out[i] : nullptr; p.ldo[i] = i < d->nseg ? d->ldo[i] : 0; }
  const size_t lds = ((size_t)(d->N + DL_COLS) * (DL_KC + 4) + 4 * (size_t)d->N) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {                                 // N = 64 rows needs 75 KB: opt in once for the maximum
    const size_t max_lds = ((size_t)(DL_MAXN + DL_COLS) * (DL_KC + 4) + 4 * (size_t)DL_MAXN) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return cape_set_error("cape_decode_linear: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(decode_linear_kernel, dim3((d->Nout + DL_COLS - 1) / DL_COLS), dim3(512), lds, as_stream(stream), p);
  CAPE_LAUNCH_CHECK("cape_decode_linear");
  return 0;
}

extern "C" int cape_decode_tail(const cape_decode_tail_desc* d, cape_stream_t stream) {
  CAPE_REQUIRE(d != nullptr, "cape_decode_tail: null descriptor");
  CAPE_REQUIRE(d->N >= 1 && d->N <= 65535 && d->L >= 1 && d->L <= 8, "cape_decode_tail: bad N / L");
  CAPE_REQUIRE(d->P4 && d->g3 && d->b3 && d->W1 && d->B1 && d->W2 && d->B2 && d->W3 && d->B3 && d->ref && d->ref_out,
               "cape_decode_tail: null pointer");
  CAPE_REQUIRE(al16(d->P4) && d->ldp % 4 == 0 && al16(d->W1) && al16(d->W2) && al16(d->W3) && al16(d->g3) && al16(d->b3),
               "cape_decode_tail: operands must be 16-byte aligned");
  if (d->Wc) CAPE_REQUIRE(d->Bc && d->cls_out && d->ncls >= 1 && d->ncls <= 6 && al16(d->Wc), "cape_decode_tail: bad class head");
  if (d->Wp) CAPE_REQUIRE(d->Bp && d->gp && d->bp && d->dim_t && d->vr && d->qpos_out && d->refin_out && al16(d->Wp) && al16(d->gp) &&
                          al16(d->bp) && al16(d->qpos_out), "cape_decode_tail: bad next-layer operands");
  if (d->hs_out) CAPE_REQUIRE(al16(d->hs_out) && d->ld_hs % 4 == 0, "cape_decode_tail: hs_out must be 16-byte aligned rows");
  DecTailP p;
  p.N = d->N; p.L = d->L; p.last = 0;
  p.P4 = d->P4; p.ldp = d->ldp; p.g3 = d->g3; p.b3 = d->b3;
  p.W1 = d->W1; p.B1 = d->B1; p.W2 = d->W2; p.B2 = d->B2; p.W3 = d->W3; p.B3 = d->B3;
  p.ref = d->ref; p.Wc = d->Wc; p.Bc = d->Bc; p.ncls = d->Wc ? d->ncls : 0;
  p.Wp = d->Wp; p.Bp = d->Bp; p.gp = d->gp; p.bp = d->bp; p.dim_t = d->dim_t; p.vr = d->vr;
  p.ref_out = d->ref_out; p.ld_ref = d->ld_ref; p.qpos_out = d->qpos_out; p.refin_out = d->refin_out;
  p.cls_out = d->Wc ? d->cls_out : nullptr; p.ld_cls = d->ld_cls; p.hs_out = d->hs_out; p.ld_hs = d->ld_hs;
  hipLaunchKernelGGL(decode_tail_kernel, dim3(d->N), dim3(512), 0, as_stream(stream), p);
  CAPE_LAUNCH_CHECK("cape_decode_tail");
  return 0;
}
