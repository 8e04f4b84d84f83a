// decode_fused.hip -- one cached autoregressive decode step of ALL decoder layers as ONE launch: one 512-thread block per
// image walks the whole step (RoomFormerV2.forward_inference -> TransformerDecoder / TransformerDecoderLayer v1 with one
// query token per image; reference models/roomformer_v2.py:481-598, deformable_transformer_v2.py:320-370, :1024-1131).
//
// Why row-split.  A step is ~20 dependent matrix-vector stages per layer.  With a launch per stage (decode_step.hip, ~75
// launches) every stage costs a launch boundary plus 2-3 dependent memory round trips: 560-760 us per step measured, 410 us
// even if every launch took the 5.5 us minimum.  Nothing in a step couples two images, so a block that owns one image can
// run the whole chain without any grid-level synchronisation; the price is that every block streams all decoder weights
// (5.2 MB per layer, 31 MB per step) through its own CU from L2 / Infinity Cache instead of the chip reading them once.
// A CU ingests ~100 GB/s, so the step is bound at ~0.3 ms by that stream -- and weight addresses do not depend on data, so
// the next stage's weight block is always requested before the current stage's reduction and barrier.
//
// Stage primitive: a wave owns 32 output rows (j = row0 + wave + 8 i) of a 256-row block and holds their 1 KB k-slices
// in registers (32 float4 = 128 VGPRs, requested together); the input vector lives in registers (4 consecutive k per
// lane); 32 dot products are reduced over the 64 lanes with the 32-shuffle transpose-reduce of decode_step.hip.
// LayerNorms are computed by every wave redundantly from the pre-norm vector in LDS (two wave reductions, no barrier).
// Single-query attention: wave = head (keys across lanes for q.k, channels across lanes for p.V); the deformable
// sampling: records by wave 0, gathers by all 8 waves (2 samples each), partial sums through LDS.
// All arithmetic is plain fp32 FMA.
#include "common.h"
#include <stddef.h>

namespace {

constexpr int C = 256, NH = 8, HD = 32;
constexpr int MAXKEYS = 1024;
constexpr int FFN = 1024;           // dim_feedforward (host-checked)

struct LevelsD { int H[4], W[4], start[4]; };
__device__ __forceinline__ int sel4d(const int (&a)[4], int l) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); }

struct DecLayerP {
  const float *w_qkv, *b_qkv, *w_qin;
  float *kc, *vc;
  const float *w_o, *b_o, *g2, *be2;
  const float *w_sq, *b_sq, *supk, *supv; const unsigned char* supm; const float *w_so, *b_so, *gs, *bes;
  const float *w_off, *b_off, *value, *w_mo, *b_mo, *g1, *be1;
  const float *w1, *b1, *w2, *b2, *g3, *be3;
  const float *m1w, *m1b, *m2w, *m2b, *m3w, *m3b;
};

struct DecStepP {
  int N, nl, step, T, P, S, L, NP, ncls, F;
  const float *emb, *qpos0, *refin0, *ref0, *vr, *dim_t;
  const float *wc, *bc, *wp, *bp, *gp, *bep;
  float *out_logits, *out_coords, *out_hs;
  long long ld_logits, ld_coords, ld_hs;
  LevelsD lv;
  DecLayerP layer[CAPE_DECODE_MAX_LAYERS];
};

struct WRows { float4 w[32]; };

// rows j = row0 + wave + 8 i clamped to rmax (a clamped row is loaded twice and its result dropped), k = k0 + 4 lane
// (scheduling fences on both sides: the loads must not be hoisted above the products that free the registers they land in --
// both weight blocks live at once is 256 VGPRs and spills -- nor sink below the reduction they are meant to overlap)
__device__ __forceinline__ void ld_rows(WRows& r, const float* __restrict__ W, int ldw, int row0, int rmax, int k0) {
  // the wave number as a scalar: row addresses are then SGPR bases + one shared lane offset (as a vector value the 32
  // clamped row addresses cost 64 VGPRs on top of the 128 they load into)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int j = min(row0 + wave + 8 * i, rmax);
    r.w[i] = *reinterpret_cast<const float4*>(W + (long long)j * ldw + k0 + 4 * lane);
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void fma_rows(float (&s)[32], const WRows& r, const float4 x, bool first) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float d = fmaf(r.w[i].x, x.x, fmaf(r.w[i].y, x.y, fmaf(r.w[i].z, x.z, r.w[i].w * x.w)));
    s[i] = first ? d : s[i] + d;
  }
}
// 32 partial dot products per lane -> one finished output in every even lane; returns the output's index i (row = row0 + wave + 8 i)
__device__ __forceinline__ int reduce32(float (&s)[32], float& v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int n = 16, o = 32; n >= 1; n >>= 1, o >>= 1) {
    const bool up = lane & o;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float keep = up ? s[2 * i + 1] : s[2 * i], send = up ? s[2 * i] : s[2 * i + 1];
      s[i] = keep + __shfl_xor(send, o, 64);
    }
  }
  v = s[0] + __shfl_xor(s[0], 1, 64);
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}

// LayerNorm of a 256-vector held in LDS: every wave computes it for itself (4 consecutive channels per lane)
__device__ __forceinline__ float4 ln256(const float* p, const float* __restrict__ g, const float* __restrict__ b) {
  const int lane = threadIdx.x & 63;
  const float4 v = *reinterpret_cast<const float4*>(p + 4 * lane);
  const float mean = wave_sum(v.x + v.y + v.z + v.w) / 256.f;
  const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
  const float rstd = rsqrtf(wave_sum(a * a + bb * bb + c * c + d * d) / 256.f + 1e-5f);
  const float4 gg = *reinterpret_cast<const float4*>(g + 4 * lane);
  const float4 be = *reinterpret_cast<const float4*>(b + 4 * lane);
  return make_float4(a * rstd * gg.x + be.x, bb * rstd * gg.y + be.y, c * rstd * gg.z + be.z, d * rstd * gg.w + be.w);
}

__device__ __forceinline__ float inv_sigmoid_d(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  return logf(x1 / x2);
}

// single-query attention of one head by one wave: nk keys in global memory (row stride 256) + an optional last key / value
// held in LDS (the token of this step).  sc = this wave's score scratch (nk + 1 floats).
__device__ __forceinline__ void attn_head(const float* qh, const float* __restrict__ K, const float* __restrict__ V, int nk,
                                          const unsigned char* __restrict__ kpm, const float* knew, const float* vnew, float* sc,
                                          float* out, float scale) {
  const int lane = threadIdx.x & 63;
  // the query is re-read from LDS per key (broadcast reads): holding it would cost 32 VGPRs next to the prefetched weight block
  float m = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const float* kr = K + (long long)j * C;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 b = *reinterpret_cast<const float4*>(kr + d);
      const float4 a = *reinterpret_cast<const float4*>(qh + d);
      s += (a.x * scale) * b.x + (a.y * scale) * b.y + (a.z * scale) * b.z + (a.w * scale) * b.w;
    }
    if (kpm && kpm[j]) s = -INFINITY;
    sc[j] = s;
    m = fmaxf(m, s);
  }
  int tot = nk;
  if (knew) {                                                    // wave-uniform
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 b = *reinterpret_cast<const float4*>(knew + d);
      const float4 a = *reinterpret_cast<const float4*>(qh + d);
      s += (a.x * scale) * b.x + (a.y * scale) * b.y + (a.z * scale) * b.z + (a.w * scale) * b.w;
    }
    if (lane == 0) sc[nk] = s;
    m = fmaxf(m, s);
    tot = nk + 1;
  }
  m = wave_max(m);
  __builtin_amdgcn_wave_barrier();
  float l = 0.f;
  for (int j = lane; j < tot; j += 64) {
    const float e = __expf(sc[j] - m);                           // a fully masked row: m = -inf -> NaN like torch
    sc[j] = e;
    l += e;
  }
  l = wave_sum(l);
  __builtin_amdgcn_wave_barrier();
  const int c = lane & 31, half = lane >> 5;
  const float* vb = V + c;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int j = half;
  for (; j + 6 < nk; j += 8) {
    const float v0 = vb[(long long)j * C], v1 = vb[(long long)(j + 2) * C];
    const float v2 = vb[(long long)(j + 4) * C], v3 = vb[(long long)(j + 6) * C];
    acc0 = fmaf(sc[j], v0, acc0); acc1 = fmaf(sc[j + 2], v1, acc1); acc2 = fmaf(sc[j + 4], v2, acc2); acc3 = fmaf(sc[j + 6], v3, acc3);
  }
  for (; j < nk; j += 2) acc0 = fmaf(sc[j], vb[(long long)j * C], acc0);
  if (vnew && half == 0) acc1 = fmaf(sc[nk], vnew[c], acc1);
  float acc = (acc0 + acc1) + (acc2 + acc3);
  acc += __shfl_xor(acc, 32, 64);
  if (lane < 32) out[c] = acc / l;
}

// LDS map (floats)
constexpr int O_XIN = 0, O_T = 256, O_T2 = 512, O_T3 = 768, O_PRE = 1024, O_Q = 1280, O_KN = 1536, O_VN = 1792, O_ATT = 2048,
              O_QPOS = 2304, O_OFFW = 2560 /*384*/, O_G = 2944, O_H = 3200 /*1024*/, O_PART = 4224 /*8 x 256*/, O_RECW = 6272 /*128 x 4*/,
              O_RECI = 6784 /*128 x 2 uint*/, O_SMALL = 7040 /*32*/, O_SC = 7072;

__global__ void __launch_bounds__(512) decode_step_kernel(const DecStepP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int scw = (max(p.T, p.P) + 1 + 3) & ~3;                  // score scratch per wave
  float* sc = sm + O_SC + wave * scw;
  float* small = sm + O_SMALL;
  WRows wr;
  float s[32];
  float v;
  const float scale = 0.17677669529663687f;                      // 32^-0.5

  // ---- layer-0 inputs
  if (t < 64) {
    *reinterpret_cast<float4*>(sm + O_XIN + 4 * t) = *reinterpret_cast<const float4*>(p.emb + (long long)n * C + 4 * t);
    *reinterpret_cast<float4*>(sm + O_QPOS + 4 * t) = *reinterpret_cast<const float4*>(p.qpos0 + 4 * t);
  }
  if (t < 2) small[t] = p.ref0[n * 2 + t];                                            // reference point of this layer
  if (t >= 64 && t < 64 + 2 * p.L) small[8 + t - 64] = p.refin0[(long long)n * p.L * 2 + (t - 64)];   // level-scaled points
  ld_rows(wr, p.layer[0].w_qkv, C, 0, 767, 0);
  __syncthreads();

  // the per-layer pointer table is read from the kernel-argument segment with scalar loads (indexing the by-value struct
  // with a runtime layer number would make the compiler copy it to scratch)
#if defined(__HIP_DEVICE_COMPILE__)
  const DecLayerP* layer_tab = (const DecLayerP*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(DecStepP, layer));
#else
  const DecLayerP* layer_tab = p.layer;          // host pass of the single-source compile: never executed
#endif
  for (int l = 0; l < p.nl; ++l) {
    const DecLayerP w = layer_tab[l];
    const bool last = l == p.nl - 1;
    // ================= A: q | k | v (folded projections), q += in_proj_q(query_pos) =================
    float4 x = *reinterpret_cast<const float4*>(sm + O_XIN + 4 * lane);
    const float4 qp = *reinterpret_cast<const float4*>(sm + O_QPOS + 4 * lane);
    float qacc = 0.f;                                             // this lane's q output (even lanes), kept across two products
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
      fma_rows(s, wr, x, true);
      ld_rows(wr, blk < 2 ? w.w_qkv : w.w_qin, C, blk < 2 ? 256 * (blk + 1) : 0, blk < 2 ? 767 : 255, 0);
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) {
        const float r = v + w.b_qkv[256 * blk + j];
        if (blk == 0) qacc = r;
        else if (blk == 1) { sm[O_KN + j] = r; w.kc[((long long)n * p.T + p.step) * C + j] = r; }
        else { sm[O_VN + j] = r; w.vc[((long long)n * p.T + p.step) * C + j] = r; }
      }
    }
    fma_rows(s, wr, qp, true);
    ld_rows(wr, w.w_o, C, 0, 255, 0);
    {
      const int i = reduce32(s, v);
      if ((lane & 1) == 0) sm[O_Q + wave + 8 * i] = qacc + v;
    }
    __syncthreads();
    // ================= B: self-attention over the cache rows 0..step-1 and this step's key =================
#ifndef LAB_NO_ATTN
    attn_head(sm + O_Q + wave * HD, w.kc + (long long)n * p.T * C + wave * HD, w.vc + (long long)n * p.T * C + wave * HD, p.step, nullptr,
              sm + O_KN + wave * HD, sm + O_VN + wave * HD, sc, sm + O_ATT + wave * HD, scale);
#endif
    __syncthreads();
    // ================= C: out_proj + residual -> p1; LN2 =================
    x = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * lane);
    fma_rows(s, wr, x, true);
    ld_rows(wr, w.w_sq, C, 0, 255, 0);
    {
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_PRE + j] = v + w.b_o[j] + sm[O_XIN + j];
    }
    __syncthreads();
    x = ln256(sm + O_PRE, w.g2, w.be2);
    if (wave == 0) *reinterpret_cast<float4*>(sm + O_T + 4 * lane) = x;
    {
      // ================= D: support cross-attention (host-checked: every layer has one) =================
      fma_rows(s, wr, x, true);
      ld_rows(wr, w.w_so, C, 0, 255, 0);
      {
        const int i = reduce32(s, v);
        const int j = wave + 8 * i;
        if ((lane & 1) == 0) sm[O_Q + j] = v + w.b_sq[j];
      }
      __syncthreads();
#ifndef LAB_NO_ATTN
      attn_head(sm + O_Q + wave * HD, w.supk + (long long)n * p.P * C + wave * HD, w.supv + (long long)n * p.P * C + wave * HD, p.P,
                w.supm ? w.supm + (long long)n * p.P : nullptr, nullptr, nullptr, sc, sm + O_ATT + wave * HD, scale);
#endif
      __syncthreads();
      x = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * lane);
      fma_rows(s, wr, x, true);
      ld_rows(wr, w.w_off, C, 0, 383, 0);
      {
        const int i = reduce32(s, v);
        const int j = wave + 8 * i;
        if ((lane & 1) == 0) sm[O_PRE + j] = v + w.b_so[j] + sm[O_T + j];
      }
      __syncthreads();
      x = ln256(sm + O_PRE, w.gs, w.bes);
      if (wave == 0) *reinterpret_cast<float4*>(sm + O_T2 + 4 * lane) = x;
    }
    // ================= E: sampling offsets | attention logits of (t + query_pos) =================
    {
      const float4 xq = make_float4(x.x + qp.x, x.y + qp.y, x.z + qp.z, x.w + qp.w);
      fma_rows(s, wr, xq, true);
      ld_rows(wr, w.w_off, C, 256, 383, 0);
      int i = reduce32(s, v);
      int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_OFFW + j] = v + w.b_off[j];
      fma_rows(s, wr, xq, true);
      ld_rows(wr, w.w_mo, C, 0, 255, 0);
      i = reduce32(s, v);
      j = 256 + wave + 8 * i;
      if ((lane & 1) == 0 && j < 384) sm[O_OFFW + j] = v + w.b_off[j];
    }
    __syncthreads();
    // ================= F: deformable sampling of the cached value projection =================
    {
      const int LP = p.L * p.NP;
      float4* rec_w = reinterpret_cast<float4*>(sm + O_RECW);
      uint2* rec_i = reinterpret_cast<uint2*>(sm + O_RECI);
#ifndef LAB_NO_MSDA
      if (wave == 0) {
        const int h = lane >> 3, i = lane & 7;
        const float* ow = sm + O_OFFW;
        float px[2], py[2], lg[2];
        int Wd[2], Hd[2], st[2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = i + 8 * q2;
          px[q2] = 0.f; py[q2] = 0.f; lg[q2] = -INFINITY; Wd[q2] = 1; Hd[q2] = 1; st[q2] = 0;
          if (j < LP) {
            const int lev = j / p.NP;
            Wd[q2] = sel4d(p.lv.W, lev); Hd[q2] = sel4d(p.lv.H, lev); st[q2] = sel4d(p.lv.start, lev);
            const float rx = small[8 + lev * 2], ry = small[8 + lev * 2 + 1];
            const float ox = ow[(h * LP + j) * 2], oy = ow[(h * LP + j) * 2 + 1];
            const float Wf = (float)Wd[q2], Hf = (float)Hd[q2];
            px[q2] = (rx + ox / Wf) * Wf - 0.5f;
            py[q2] = (ry + oy / Hf) * Hf - 0.5f;
            lg[q2] = ow[NH * LP * 2 + h * LP + j];
          }
        }
        float mx = fmaxf(lg[0], lg[1]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        const float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx);
        float se = e0 + e1;
        se += __shfl_xor(se, 1, 64); se += __shfl_xor(se, 2, 64); se += __shfl_xor(se, 4, 64);
        const float inv = 1.f / se;
        const float aw[2] = {e0 * inv, e1 * inv};
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = i + 8 * q2;
          const float xf = floorf(px[q2]), yf = floorf(py[q2]);
          const float fx = px[q2] - xf, fy = py[q2] - yf;
          const int x0 = (int)xf, y0 = (int)yf;
          const bool live = j < LP;
          const bool xa = x0 >= 0 && x0 < Wd[q2], xb = x0 + 1 >= 0 && x0 + 1 < Wd[q2];
          const bool ya = y0 >= 0 && y0 < Hd[q2], yb = y0 + 1 >= 0 && y0 + 1 < Hd[q2];
          const int base = st[q2] + y0 * Wd[q2] + x0;
          const unsigned cl = (unsigned)(p.S - 1);
          const bool v00 = live && ya && xa, v01 = live && ya && xb, v10 = live && yb && xa, v11 = live && yb && xb;
          const unsigned i00 = v00 ? (unsigned)base : cl, i01 = v01 ? (unsigned)(base + 1) : cl;
          const unsigned i10 = v10 ? (unsigned)(base + Wd[q2]) : cl, i11 = v11 ? (unsigned)(base + Wd[q2] + 1) : cl;
          rec_w[h * 16 + j] = make_float4(v00 ? aw[q2] * (1.f - fx) * (1.f - fy) : 0.f, v01 ? aw[q2] * fx * (1.f - fy) : 0.f,
                                          v10 ? aw[q2] * (1.f - fx) * fy : 0.f, v11 ? aw[q2] * fx * fy : 0.f);
          rec_i[h * 16 + j] = make_uint2(i00 | (i01 << 16), i10 | (i11 << 16));
        }
      }
      __syncthreads();
      {
        const int h = lane >> 3, i = lane & 7;
        const float* vb = w.value + (long long)n * p.S * C + h * HD + i * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g4[2][4];
        uint2 ids[2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = 2 * wave + q2;                            // LP <= 16: samples beyond LP carry zero weights and clamped ids
          ids[q2] = rec_i[h * 16 + j];
          g4[q2][0] = *reinterpret_cast<const float4*>(vb + (ids[q2].x & 0xFFFFu) * (unsigned)C);
          g4[q2][1] = *reinterpret_cast<const float4*>(vb + (ids[q2].x >> 16) * (unsigned)C);
          g4[q2][2] = *reinterpret_cast<const float4*>(vb + (ids[q2].y & 0xFFFFu) * (unsigned)C);
          g4[q2][3] = *reinterpret_cast<const float4*>(vb + (ids[q2].y >> 16) * (unsigned)C);
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const float4 ww = rec_w[h * 16 + 2 * wave + q2];
          acc.x += ww.x * g4[q2][0].x + ww.y * g4[q2][1].x + ww.z * g4[q2][2].x + ww.w * g4[q2][3].x;
          acc.y += ww.x * g4[q2][0].y + ww.y * g4[q2][1].y + ww.z * g4[q2][2].y + ww.w * g4[q2][3].y;
          acc.z += ww.x * g4[q2][0].z + ww.y * g4[q2][1].z + ww.z * g4[q2][2].z + ww.w * g4[q2][3].z;
          acc.w += ww.x * g4[q2][0].w + ww.y * g4[q2][1].w + ww.z * g4[q2][2].w + ww.w * g4[q2][3].w;
        }
        *reinterpret_cast<float4*>(sm + O_PART + wave * C + h * HD + i * 4) = acc;
      }
#endif
      __syncthreads();
      if (t < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += sm[O_PART + k * C + t];
        sm[O_G + t] = a;
      }
      __syncthreads();
    }
    // ================= G: output_proj + residual -> p3; LN1 =================
    x = *reinterpret_cast<const float4*>(sm + O_G + 4 * lane);
    fma_rows(s, wr, x, true);
    ld_rows(wr, w.w1, C, 0, FFN - 1, 0);
    {
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_PRE + j] = v + w.b_mo[j] + sm[O_T2 + j];
    }
    __syncthreads();
    x = ln256(sm + O_PRE, w.g1, w.be1);
    if (wave == 0) *reinterpret_cast<float4*>(sm + O_T3 + 4 * lane) = x;
    // ================= H: feed-forward =================
    constexpr int nb1 = FFN / 256;
#pragma unroll
    for (int blk = 0; blk < nb1; ++blk) {
      fma_rows(s, wr, x, true);
      if (blk + 1 < nb1) ld_rows(wr, w.w1, C, 256 * (blk + 1), FFN - 1, 0);
      else ld_rows(wr, w.w2, FFN, 0, 255, 0);
      const int i = reduce32(s, v);
      const int j = 256 * blk + wave + 8 * i;
      if ((lane & 1) == 0) sm[O_H + j] = fmaxf(v + w.b1[j], 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < nb1; ++blk) {
      const float4 hx = *reinterpret_cast<const float4*>(sm + O_H + 256 * blk + 4 * lane);
      fma_rows(s, wr, hx, blk == 0);
      if (blk + 1 < nb1) ld_rows(wr, w.w2, FFN, 0, 255, 256 * (blk + 1));
      else ld_rows(wr, w.m1w, C, 0, 255, 0);
    }
    {
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_PRE + j] = v + w.b2[j] + sm[O_T3 + j];
    }
    __syncthreads();
    // ================= I: tail -- LN3, class head, coords MLP, refinement, next query position embedding =================
    x = ln256(sm + O_PRE, w.g3, w.be3);
    if (wave == 0) {
      *reinterpret_cast<float4*>(sm + O_XIN + 4 * lane) = x;                           // next layer's input (and its residual)
      if (last) *reinterpret_cast<float4*>(p.out_hs + (long long)n * p.ld_hs + 4 * lane) = x;
    }
    if (last && wave < p.ncls) {
      const float4 wc = *reinterpret_cast<const float4*>(p.wc + (long long)wave * C + 4 * lane);
      const float c = wave_sum(fmaf(wc.x, x.x, fmaf(wc.y, x.y, fmaf(wc.z, x.z, wc.w * x.w))));
      if (lane == 0) p.out_logits[(long long)n * p.ld_logits + wave] = c + p.bc[wave];
    }
    fma_rows(s, wr, x, true);
    ld_rows(wr, w.m2w, C, 0, 255, 0);
    {
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_G + j] = fmaxf(v + w.m1b[j], 0.f);
    }
    __syncthreads();
    x = *reinterpret_cast<const float4*>(sm + O_G + 4 * lane);
    fma_rows(s, wr, x, true);
    ld_rows(wr, p.wp, C, 0, 255, 0);                             // (requested on the last layer too: keeps the code branch-free)
    {
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_ATT + j] = fmaxf(v + w.m2b[j], 0.f);
    }
    __syncthreads();
    if (wave < 2) {
      const float4 hh = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * lane);
      const float4 w3 = *reinterpret_cast<const float4*>(w.m3w + (long long)wave * C + 4 * lane);
      const float dlt = wave_sum(fmaf(w3.x, hh.x, fmaf(w3.y, hh.y, fmaf(w3.z, hh.z, w3.w * hh.w))));
      if (lane == 0) {
        const float z = dlt + w.m3b[wave] + inv_sigmoid_d(small[wave]);
        const float r = 1.f / (1.f + expf(-z));
        small[4 + wave] = r;
        if (last) p.out_coords[(long long)n * p.ld_coords + wave] = r;
      }
    }
    __syncthreads();
    if (last) break;
    const float rx = small[4], ry = small[5];
    __syncthreads();                                              // everyone has read the refined point before it is republished
    if (t < 2) small[t] = t ? ry : rx;                                                  // reference point of the next layer
    if (t >= 64 && t < 64 + 2 * p.L) {
      const int k = t - 64;
      small[8 + k] = ((k & 1) ? ry : rx) * p.vr[(long long)n * p.L * 2 + k];
    }
    {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * lane + i, k = c & 127;
        const float a = ((c >> 7) ? ry : rx) * 6.283185307179586f / p.dim_t[k];
        e[i] = (k & 1) ? cosf(a) : sinf(a);
      }
      fma_rows(s, wr, make_float4(e[0], e[1], e[2], e[3]), true);
      ld_rows(wr, layer_tab[l + 1].w_qkv, C, 0, 767, 0);
      const int i = reduce32(s, v);
      const int j = wave + 8 * i;
      if ((lane & 1) == 0) sm[O_PRE + j] = v + p.bp[j];
    }
    __syncthreads();
    {
      const float4 qn = ln256(sm + O_PRE, p.gp, p.bep);          // O_QPOS' old value was read at the top of this layer
      if (wave == 0) *reinterpret_cast<float4*>(sm + O_QPOS + 4 * lane) = qn;
    }
    __syncthreads();
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int cape_decode_step(const cape_decode_step_desc* d, cape_stream_t stream) {
  CAPE_REQUIRE(d != nullptr, "cape_decode_step: null descriptor");
  CAPE_REQUIRE(d->N >= 1 && d->N <= 65535, "cape_decode_step: N=%d images", d->N);
  CAPE_REQUIRE(d->n_layers >= 1 && d->n_layers <= CAPE_DECODE_MAX_LAYERS, "cape_decode_step: %d layers, at most %d", d->n_layers, CAPE_DECODE_MAX_LAYERS);
  CAPE_REQUIRE(d->T >= 1 && d->T <= MAXKEYS && d->step >= 0 && d->step < d->T, "cape_decode_step: step %d outside the cache of %d rows (max %d)", d->step, d->T, MAXKEYS);
  CAPE_REQUIRE(d->P >= 0 && d->P <= MAXKEYS, "cape_decode_step: P=%d support keys", d->P);
  CAPE_REQUIRE(d->L >= 1 && d->L <= 4 && d->n_points >= 1 && d->L * d->n_points == 16, "cape_decode_step: L=%d levels x %d points must be 16", d->L, d->n_points);
  CAPE_REQUIRE(d->S >= 1 && d->S < 65535, "cape_decode_step: S=%d memory tokens (ids are 16 bit)", d->S);
  CAPE_REQUIRE(d->ffn_dim == FFN, "cape_decode_step: ffn_dim=%d, the kernel is built for %d", d->ffn_dim, FFN);
  CAPE_REQUIRE(d->ncls >= 1 && d->ncls <= 8, "cape_decode_step: ncls=%d", d->ncls);
  CAPE_REQUIRE(d->emb && d->qpos0 && d->refin0 && d->ref0 && d->vr && d->dim_t && d->class_w && d->class_b && d->out_logits && d->out_coords && d->out_hs,
               "cape_decode_step: null pointer");
  CAPE_REQUIRE(al16(d->emb) && al16(d->qpos0) && al16(d->class_w) && al16(d->out_hs) && d->ld_hs % 4 == 0, "cape_decode_step: rows must be 16-byte aligned");
  CAPE_REQUIRE(d->pos_w && d->pos_b && d->pos_gamma && d->pos_beta && al16(d->pos_w) && al16(d->pos_gamma) && al16(d->pos_beta),
               "cape_decode_step: pos_trans operands");
  long long tot = 0;
  for (int l = 0; l < d->L; ++l) {
    CAPE_REQUIRE(d->shapes[2 * l] > 0 && d->shapes[2 * l + 1] > 0 && d->level_start[l] == tot, "cape_decode_step: level %d shape/start inconsistent", l);
    tot += (long long)d->shapes[2 * l] * d->shapes[2 * l + 1];
  }
  CAPE_REQUIRE(tot == d->S, "cape_decode_step: sum(H*W)=%lld != S=%d", tot, d->S);
  DecStepP p;
  p.N = d->N; p.nl = d->n_layers; p.step = d->step; p.T = d->T; p.P = d->P; p.S = d->S; p.L = d->L; p.NP = d->n_points; p.ncls = d->ncls; p.F = d->ffn_dim;
  p.emb = d->emb; p.qpos0 = d->qpos0; p.refin0 = d->refin0; p.ref0 = d->ref0; p.vr = d->vr; p.dim_t = d->dim_t;
  p.wc = d->class_w; p.bc = d->class_b; p.wp = d->pos_w; p.bp = d->pos_b; p.gp = d->pos_gamma; p.bep = d->pos_beta;
  p.out_logits = d->out_logits; p.out_coords = d->out_coords; p.out_hs = d->out_hs;
  p.ld_logits = d->ld_logits; p.ld_coords = d->ld_coords; p.ld_hs = d->ld_hs;
  for (int l = 0; l < 4; ++l) {
    p.lv.H[l] = l < d->L ? d->shapes[2 * l] : 1; p.lv.W[l] = l < d->L ? d->shapes[2 * l + 1] : 1; p.lv.start[l] = l < d->L ? d->level_start[l] : 0;
  }
  for (int l = 0; l < d->n_layers; ++l) {
    const cape_decode_layer_desc& s = d->layers[l];
    const float* req[] = {s.w_qkv, s.b_qkv, s.w_qin, s.k_cache, s.v_cache, s.w_o, s.b_o, s.ln2_g, s.ln2_b, s.w_off, s.b_off, s.value, s.w_mo, s.b_mo,
                          s.ln1_g, s.ln1_b, s.w1, s.b1, s.w2, s.b2, s.ln3_g, s.ln3_b, s.m1w, s.m1b, s.m2w, s.m2b, s.m3w, s.m3b};
    for (const float* q : req) CAPE_REQUIRE(q != nullptr && (reinterpret_cast<uintptr_t>(q) & 3) == 0, "cape_decode_step: layer %d: null / misaligned pointer", l);
    const float* vec[] = {s.w_qkv, s.w_qin, s.k_cache, s.v_cache, s.w_o, s.ln2_g, s.ln2_b, s.w_off, s.value, s.w_mo, s.ln1_g, s.ln1_b, s.w1, s.w2, s.ln3_g, s.ln3_b,
                          s.m1w, s.m2w, s.m3w};
    for (const float* q : vec) CAPE_REQUIRE(al16(q), "cape_decode_step: layer %d: matrices, caches and LayerNorm vectors must be 16-byte aligned", l);
    {
      CAPE_REQUIRE(s.w_sq && d->P >= 1 && s.b_sq && s.sup_k && s.sup_v && s.w_so && s.b_so && s.lns_g && s.lns_b && al16(s.w_sq) && al16(s.sup_k) && al16(s.sup_v) &&
                   al16(s.w_so) && al16(s.lns_g) && al16(s.lns_b), "cape_decode_step: layer %d: support attention operands", l);
    }
    DecLayerP& q = p.layer[l];
    q.w_qkv = s.w_qkv; q.b_qkv = s.b_qkv; q.w_qin = s.w_qin; q.kc = s.k_cache; q.vc = s.v_cache; q.w_o = s.w_o; q.b_o = s.b_o; q.g2 = s.ln2_g; q.be2 = s.ln2_b;
    q.w_sq = s.w_sq; q.b_sq = s.b_sq; q.supk = s.sup_k; q.supv = s.sup_v; q.supm = s.sup_mask; q.w_so = s.w_so; q.b_so = s.b_so; q.gs = s.lns_g; q.bes = s.lns_b;
    q.w_off = s.w_off; q.b_off = s.b_off; q.value = s.value; q.w_mo = s.w_mo; q.b_mo = s.b_mo; q.g1 = s.ln1_g; q.be1 = s.ln1_b;
    q.w1 = s.w1; q.b1 = s.b1; q.w2 = s.w2; q.b2 = s.b2; q.g3 = s.ln3_g; q.be3 = s.ln3_b;
    q.m1w = s.m1w; q.m1b = s.m1b; q.m2w = s.m2w; q.m2b = s.m2b; q.m3w = s.m3w; q.m3b = s.m3b;
  }
  const int scw = ((d->T > d->P ? d->T : d->P) + 1 + 3) & ~3;
  const size_t lds = (size_t)(O_SC + 8 * scw) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t max_lds = (size_t)(O_SC + 8 * ((MAXKEYS + 4) & ~3)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return cape_set_error("cape_decode_step: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(decode_step_kernel, dim3(d->N), dim3(512), lds, as_stream(stream), p);
  CAPE_LAUNCH_CHECK("cape_decode_step");
  return 0;
}
