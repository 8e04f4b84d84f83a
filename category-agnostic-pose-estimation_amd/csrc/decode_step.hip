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

// LayerNorm of rows held in LDS, in place: wave w takes rows w, w+8, ... (C <= 1024 handled in chunks of 256 by the callers
// that need it: here C == row length in LDS == 256)
__device__ __forceinline__ void ln_rows_lds(float* rows, int ld, int N, const float* gamma, const float* beta, const float* add,
                                            long long ld_add) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < N; r += 8) {
    float4 v = *reinterpret_cast<float4*>(&rows[r * ld + 4 * lane]);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) / 256.f;
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) / 256.f + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * lane);
    const float4 be = *reinterpret_cast<const float4*>(beta + 4 * lane);
    v = make_float4(a * rstd * g.x + be.x, b * rstd * g.y + be.y, c * rstd * g.z + be.z, d * rstd * g.w + be.w);
    if (add) {
      const float4 q = *reinterpret_cast<const float4*>(add + (long long)r * ld_add + 4 * lane);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(&rows[r * ld + 4 * lane]) = v;
  }
}

// LDS: xs[N][260] | ws[8][260] | rs[N][260] (normalised residual rows, only with res_gamma).
// Every global operand of the first k-chunk is requested before the first wait: the kernel's critical path is ONE memory
// round trip (+ one per extra k-chunk / second product), then LDS-resident LayerNorms, then the dot products.
__global__ void __launch_bounds__(512) decode_linear_kernel(const DecLinP p) {
  extern __shared__ __attribute__((aligned(16))) float dl_lds[];
  constexpr int LD = DL_KC + 4;
  float* xs = dl_lds;
  float* ws = xs + p.N * LD;
  float* rs = ws + DL_COLS * LD;
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * DL_COLS;
  const int col = t & (DL_COLS - 1), row = t >> 3;
  float acc = 0.f;
  const int npass = (p.X2 && n0 < p.n2) ? 2 : 1;
  bool res_staged = false;
  for (int pass = 0; pass < npass; ++pass) {
    const float* X = pass ? p.X2 : p.X;
    const long long ldx = pass ? p.ldx2 : p.ldx;
    const float* W = pass ? p.W2 : p.W;
    const long long ldw = pass ? p.ldw2 : p.ldw;
    const int K = pass ? p.K2 : p.K;
    const bool ln = !pass && p.in_gamma;                       // host-checked: LN-on-load only with K == 256
    for (int k0 = 0; k0 < K; k0 += DL_KC) {
      const int kc = min(DL_KC, K - k0), kq = kc >> 2;
      if (pass || k0) __syncthreads();
      for (int i = t; i < DL_COLS * kq; i += 512) {            // weights first: they are the bytes that come from far away
        const int r = i / kq, c = (i - r * kq) * 4;
        const int n = min(n0 + r, p.Nout - 1);
        *reinterpret_cast<float4*>(&ws[r * LD + c]) = *reinterpret_cast<const float4*>(W + (long long)n * ldw + k0 + c);
      }
      for (int i = t; i < p.N * kq; i += 512) {
        const int r = i / kq, c = (i - r * kq) * 4;
        float4 v = *reinterpret_cast<const float4*>(X + (long long)r * ldx + k0 + c);
        if (!ln && !pass && p.in_add) {
          const float4 a = *reinterpret_cast<const float4*>(p.in_add + (long long)r * p.ld_add + k0 + c);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(&xs[r * LD + c]) = v;
      }
      if (!res_staged && p.res_gamma) {                        // residual rows (width Nout == 256, host-checked) for their LayerNorm
        for (int i = t; i < p.N * 64; i += 512) {
          const int r = i >> 6, c = (i & 63) * 4;
          *reinterpret_cast<float4*>(&rs[r * LD + c]) = *reinterpret_cast<const float4*>(p.R + (long long)r * p.ldr + c);
        }
      }
      __syncthreads();
      if (ln || (!res_staged && p.res_gamma)) {
        if (ln) ln_rows_lds(xs, LD, p.N, p.in_gamma, p.in_beta, p.in_add, p.ld_add);
        if (!res_staged && p.res_gamma) ln_rows_lds(rs, LD, p.N, p.res_gamma, p.res_beta, nullptr, 0);
        res_staged = true;
        __syncthreads();
      }
      if (row < p.N) {
        const float* xr = &xs[row * LD];
        const float* wr = &ws[col * LD];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int k = 0; k < kc; k += 4) {
          const float4 a = *reinterpret_cast<const float4*>(xr + k);
          const float4 b = *reinterpret_cast<const float4*>(wr + k);
          a0 = fmaf(a.x, b.x, a0); a1 = fmaf(a.y, b.y, a1); a2 = fmaf(a.z, b.z, a2); a3 = fmaf(a.w, b.w, a3);
        }
        acc += (a0 + a1) + (a2 + a3);
      }
    }
  }
  const int n = n0 + col;
  if (row < p.N && n < p.Nout) {
    float v = acc + (p.bias ? p.bias[n] : 0.f);
    if (p.R) v += p.res_gamma ? rs[row * LD + n] : p.R[(long long)row * p.ldr + n];
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

// ---- tail building blocks.  A wave owns outputs j = wave + 8 i, i = 0..31 of a 256 x 256 product; x lives in registers
// (4 consecutive k per lane), so does the wave's 32 x 1 KB weight block (128 VGPRs): all of it is requested at once.
struct WRows { float4 w[32]; };

__device__ __forceinline__ void load_rows256(WRows& r, const float* W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 32; ++i) r.w[i] = *reinterpret_cast<const float4*>(W + (long long)(wave + 8 * i) * 256 + 4 * lane);
}

// 32 dot products per wave reduced over the 64 lanes with 32 shuffles instead of 32 x 6: each round a lane keeps half of its
// values (which half = one bit of the lane id) and hands the other half to the partner that keeps them, so after the rounds
// 32..2 every lane holds ONE output summed over its 32-lane class and a last exchange with lane ^ 1 finishes it.
// The output a lane ends up with is i = bits (5 4 3 2 1) of the lane id read as (b5 + 2 b4 + 4 b3 + 8 b2 + 16 b1).
__device__ __forceinline__ void gemv256(const WRows& r, const float* B, const float4 x, float* y, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) s[i] = fmaf(r.w[i].x, x.x, fmaf(r.w[i].y, x.y, fmaf(r.w[i].z, x.z, r.w[i].w * x.w)));
#pragma unroll
  for (int n = 16, o = 32; n >= 1; n >>= 1, o >>= 1) {
    const bool up = lane & o;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float keep = up ? s[2 * i + 1] : s[2 * i], send = up ? s[2 * i] : s[2 * i + 1];
      s[i] = keep + __shfl_xor(send, o, 64);
    }
  }
  const float v = s[0] + __shfl_xor(s[0], 1, 64);
  if ((lane & 1) == 0) {
    const int i = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
    const int j = wave + 8 * i;
    const float t = v + (B ? B[j] : 0.f);
    y[j] = relu ? fmaxf(t, 0.f) : t;
  }
}

// one output row per wave (class head, last MLP layer): row already in registers
__device__ __forceinline__ float dot_row(const float4 w, const float4 x) {
  return wave_sum(fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, w.w * x.w))));
}

__global__ void __launch_bounds__(512) decode_tail_kernel(const DecTailP p) {
  __shared__ __attribute__((aligned(16))) float buf[2][256];
  __shared__ float small[8];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // everything that does not depend on data is requested up front: the row, the first MLP layer's weight block, the
  // last MLP layer's two rows and the class-head rows (one row per wave)
  float4 x = *reinterpret_cast<const float4*>(p.P4 + (long long)n * p.ldp + 4 * lane);
  WRows wr;
  load_rows256(wr, p.W1);
  const float4 w3 = *reinterpret_cast<const float4*>(p.W3 + (long long)(wave & 1) * 256 + 4 * lane);
  const float4 wc = p.Wc ? *reinterpret_cast<const float4*>(p.Wc + (long long)min(wave, p.ncls - 1) * 256 + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  // LN3 of this row, redundantly in every wave (no barrier): x = 4 consecutive channels per lane
  {
    const float mean = wave_sum(x.x + x.y + x.z + x.w) / 256.f;
    const float a = x.x - mean, b = x.y - mean, c = x.z - mean, d = x.w - mean;
    const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) / 256.f + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(p.g3 + 4 * lane);
    const float4 be = *reinterpret_cast<const float4*>(p.b3 + 4 * lane);
    x = make_float4(a * rstd * g.x + be.x, b * rstd * g.y + be.y, c * rstd * g.z + be.z, d * rstd * g.w + be.w);
  }
  if (p.hs_out && wave == 0) *reinterpret_cast<float4*>(p.hs_out + (long long)n * p.ld_hs + 4 * lane) = x;
  if (p.Wc && wave < p.ncls) {                                              // class head reads the layer output
    const float c = dot_row(wc, x);
    if (lane == 0) small[2 + wave] = c + p.Bc[wave];
  }
  gemv256(wr, p.B1, x, buf[0], true);
  load_rows256(wr, p.W2);                                                  // in flight across the barrier
  __syncthreads();
  float4 h = *reinterpret_cast<const float4*>(&buf[0][4 * lane]);
  gemv256(wr, p.B2, h, buf[1], true);
  if (p.Wp) load_rows256(wr, p.Wp);                                        // next layer's pos_trans block, behind the refinement
  __syncthreads();
  h = *reinterpret_cast<const float4*>(&buf[1][4 * lane]);
  if (wave < 2) {
    const float dlt = dot_row(w3, h);
    if (lane == 0) {
      const float z = dlt + p.B3[wave] + inv_sigmoid_f(p.ref[n * 2 + wave]);
      const float r = 1.f / (1.f + expf(-z));
      small[wave] = r;
      p.ref_out[(long long)n * p.ld_ref + wave] = r;
    }
  }
  __syncthreads();
  if (p.cls_out && threadIdx.x < p.ncls) p.cls_out[(long long)n * p.ld_cls + threadIdx.x] = small[2 + threadIdx.x];
  if (!p.Wp) return;
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
  gemv256(wr, p.Bp, make_float4(e[0], e[1], e[2], e[3]), buf[0], false);
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
  if (d->in_gamma) CAPE_REQUIRE(d->K == 256 && al16(d->in_gamma) && al16(d->in_beta), "cape_decode_linear: LN-on-load needs K == 256 (the model width)");
  if (d->in_add) CAPE_REQUIRE(al16(d->in_add) && d->ld_add % 4 == 0, "cape_decode_linear: in_add must be 16-byte aligned rows");
  if (d->res_gamma) CAPE_REQUIRE(d->R && d->Nout == 256 && al16(d->R) && d->ldr % 4 == 0 && al16(d->res_gamma) && al16(d->res_beta),
                                 "cape_decode_linear: a normalised residual needs aligned rows of width Nout == 256");
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
  const size_t lds = ((size_t)(2 * d->N + DL_COLS) * (DL_KC + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {                                 // N = 64 rows needs 75 KB: opt in once for the maximum
    const size_t max_lds = ((size_t)(2 * DL_MAXN + DL_COLS) * (DL_KC + 4)) * sizeof(float);
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
