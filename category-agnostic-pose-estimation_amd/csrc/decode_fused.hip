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
// Stage primitive: a 256-row block of a weight matrix is consumed as four 64-row pieces (a wave owns 8 rows of a piece and
// holds their 1 KB k-slices as 8 float4); the four piece buffers form a ring -- as soon as a piece has been multiplied, the
// same piece of the NEXT block of the step's static schedule is requested into it -- so ~200 KB per CU stay in flight across
// stage boundaries.  The input vector lives in registers (4 consecutive k per lane); the 8 dot products of a piece are reduced
// over the 64 lanes by a transpose-reduce on the VALU (v_permlane32_swap, v_permlane16_swap, DPP row mirror, 8-lane DPP sum).
// LayerNorms are computed by every wave redundantly from the pre-norm vector in LDS (two wave reductions, no barrier).
// Single-query attention: wave = head (keys across lanes for q.k, channels across lanes for p.V); the deformable
// sampling: records by wave 0, gathers by all 8 waves (2 samples each), partial sums through LDS.
// Measured: 0.54 ms per step for 2 ... 32 images, 0.63 ms for 128 (GEMV chain alone 0.43 ms = 69 % of the stream rate).
// LAB_NO_ATTN / LAB_NO_MSDA (compile-time) cut the attention / sampling stages out for timing the chain alone.
// All arithmetic is plain fp32 FMA.
#include "common.h"
#include <stddef.h>

namespace {

constexpr int C = 256, NH = 8, HD = 32;
constexpr int MAXKEYS = 1024;
constexpr int FFN = 1024;           // dim_feedforward (host-checked)
constexpr int NW = 8;               // waves per block
constexpr int RP = 8 * NW;          // rows of a ring piece (8 per wave)
constexpr int NQ = 256 / RP;        // ring depth = pieces of a 256-row block

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

// ---- the weight stream.  A 256-row block of a matrix is consumed as four QUARTERS of 64 rows (a wave owns 8 rows of a
// quarter: j = row0 + 64 c + wave + 8 i, and holds their 1 KB k-slices as 8 float4 = 32 VGPRs).  The four quarter buffers
// form a ring: as soon as a quarter has been multiplied, the same quarter of the NEXT block of the step's static schedule
// is requested into it, so ~192-256 KB per CU are always in flight and the memory pipe never drains at a stage boundary
// (tools/lab/cu_ingest.hip: a CU ingests ~110 GB/s when the stream is continuous; with one 256-row request per stage and a
// barrier behind it the same kernel ran at half that).
// ---- cross-lane helpers on the VALU (DPP inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows -- new on
// gfx950): a ds_bpermute shuffle is an LDS-crossbar round trip, and a step is a chain of ~900 dependent reductions
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// a + b after swapping the upper half of a with the lower half of b: lanes < 32 get a summed over both halves, lanes >= 32 b
__device__ __forceinline__ float swap32_add(float a, float b) {
  // inline asm: with the builtin, hipcc 7.2 adds result 0 to itself (r.x + r.x) when both results feed one add
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// the same one level down: even 16-lane rows get a summed over the row pair, odd rows b
__device__ __forceinline__ float swap16_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float sum8f(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }   // quad xor 1, xor 2, half-row mirror
__device__ __forceinline__ float wsum(float v) {
  v = sum8f(v);
  v += dppf<0x140>(v);                                           // row mirror: 16-lane row sums
  v = swap16_add(v, v);
  return swap32_add(v, v);
}
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v)); v = fmaxf(v, dppf<0x140>(v));
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  v = fmaxf(a, b);
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

struct W8 { float4 w[8]; };
struct Blk { const float* W; int ldw, row0, rmax, k0, nq; };       // rows row0..row0+255 clamped to rmax, k-slice k0..k0+255, nq pieces
__device__ __forceinline__ Blk blk(const float* W, int ldw, int row0, int rmax, int k0, int nq = NQ) { return Blk{W, ldw, row0, rmax, k0, nq}; }

// (scheduling fences on both sides: the loads must not be hoisted above the products that free the registers they land in,
// nor sink below the reduction they are meant to overlap; the wave number is a scalar so that row addresses are SGPR bases
// plus one shared lane offset)
__device__ __forceinline__ void ld8(W8& r, const Blk& b, int c) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_amdgcn_sched_barrier(0);
  if (c < b.nq) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = min(b.row0 + RP * c + wave + NW * i, b.rmax);
      const float4* rowp = reinterpret_cast<const float4*>(b.W + (long long)j * b.ldw + b.k0);      // uniform: SGPR base
      r.w[i] = rowp[(unsigned)lane];                                                                  // + 32-bit lane offset
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void fma8(float (&s)[8], const W8& r, const float4 x, bool first) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = fmaf(r.w[i].x, x.x, fmaf(r.w[i].y, x.y, fmaf(r.w[i].z, x.z, r.w[i].w * x.w)));
    s[i] = first ? d : s[i] + d;
  }
  // pin the products HERE: left alone, the optimiser sinks them to their first use (the reduction, or the next block's
  // accumulate), which keeps this piece's weights live next to the ones just requested into "its" registers -- two rings
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(s[i]));
}
// 8 partial dot products per lane -> one finished output in the lanes with (lane & 7) == 0: three transpose rounds (each lane
// keeps half of its values, which half = one bit of the lane id), then the sum over the 8-lane group.
// The output such a lane ends up with is i = b5 + 2 b4 + 4 b3 of its lane id.
__device__ __forceinline__ float reduce8(float (&s)[8]) {
  const int lane = threadIdx.x & 63;
  const float t0 = swap32_add(s[0], s[1]), t1 = swap32_add(s[2], s[3]), t2 = swap32_add(s[4], s[5]), t3 = swap32_add(s[6], s[7]);
  const float u0 = swap16_add(t0, t1), u1 = swap16_add(t2, t3);
  const bool up = lane & 8;
  const float keep = up ? u1 : u0, send = up ? u0 : u1;
  return sum8f(keep + dppf<0x140>(send));                        // row mirror pairs lane i with 15 - i: the other 8-lane group
}

// one 256-row block against x: out(j, sum + bias[j]) for its rows j < nvalid; requests `nxt` quarter by quarter.
// The bias values are requested BEFORE the first quarter of `nxt`: loads return in order, an epilogue that waited for a
// bias issued behind the prefetch would wait for the whole prefetch.
template <class Epi>
__device__ __forceinline__ void gemv256(W8 (&Q)[NQ], const float4 x, const float* __restrict__ bias, int r0, int nvalid, int nq, const Blk& nxt,
                                        const int i8, Epi epi) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float b[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) b[c] = bias[min(r0 + RP * c + wave + NW * i8, nvalid - 1)];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    float s[8];
    if (c < nq) fma8(s, Q[c], x, true);
    ld8(Q[c], nxt, c);
    if (c < nq) {
      const float v = reduce8(s);
      const int j = r0 + RP * c + wave + NW * i8;
      if ((lane & 7) == 0 && j < nvalid) epi(j, v + b[c]);
    }
  }
}
// accumulating form (several blocks add into the same 256 outputs: the folded query + its position term, the four k-chunks
// of linear2): products only, `acc` stays in registers until finish256
__device__ __forceinline__ void gemv256_acc(W8 (&Q)[NQ], const float4 x, float (&acc)[NQ][8], bool first, const Blk& nxt) {
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    fma8(acc[c], Q[c], x, first);
    ld8(Q[c], nxt, c);
  }
}
template <class Epi>
__device__ __forceinline__ void finish256(float (&acc)[NQ][8], const float (&b)[NQ], const int i8, Epi epi) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const float v = reduce8(acc[c]);
    if ((lane & 7) == 0) epi(RP * c + wave + NW * i8, v + b[c]);
  }
}

// LayerNorm of a 256-vector held in LDS: every wave computes it for itself (4 consecutive channels per lane)
__device__ __forceinline__ float4 ln256(const float* p, const float* __restrict__ g, const float* __restrict__ b) {
  const int lane = threadIdx.x & 63;
  const float4 v = *reinterpret_cast<const float4*>(p + 4 * lane);
  const float mean = wsum(v.x + v.y + v.z + v.w) / 256.f;
  const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
  const float rstd = rsqrtf(wsum(a * a + bb * bb + c * c + d * d) / 256.f + 1e-5f);
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
  m = wmax(m);
  __builtin_amdgcn_wave_barrier();
  float l = 0.f;
  for (int j = lane; j < tot; j += 64) {
    const float e = __expf(sc[j] - m);                           // a fully masked row: m = -inf -> NaN like torch
    sc[j] = e;
    l += e;
  }
  l = wsum(l);
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

__global__ void __launch_bounds__(64 * NW) decode_step_kernel(const DecStepP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int i8 = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);
  const int scw = (max(p.T, p.P) + 1 + 3) & ~3;                  // score scratch per wave
  float* sc = sm + O_SC + wave * scw;
  float* small = sm + O_SMALL;
  W8 Q[NQ];
  const float scale = 0.17677669529663687f;                      // 32^-0.5
  // the per-layer pointer table is read from the kernel-argument segment with scalar loads (indexing the by-value struct
  // with a runtime layer number would make the compiler copy it to scratch)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) DecLayerP* LayerTab;      // constant address space: s_load, values in SGPRs
  const LayerTab layer_tab = (LayerTab)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                        offsetof(DecStepP, layer));
#else
  const DecLayerP* layer_tab = p.layer;          // host pass of the single-source compile: never executed
#endif
#define LW(f) (layer_tab[l].f)

  // ---- layer-0 inputs
  if (t < 64) {
    *reinterpret_cast<float4*>(sm + O_XIN + 4 * t) = *reinterpret_cast<const float4*>(p.emb + (long long)n * C + 4 * t);
    *reinterpret_cast<float4*>(sm + O_QPOS + 4 * t) = *reinterpret_cast<const float4*>(p.qpos0 + 4 * t);
  }
  if (t < 2) small[t] = p.ref0[n * 2 + t];                                            // reference point of this layer
  if (t >= 64 && t < 64 + 2 * p.L) small[8 + t - 64] = p.refin0[(long long)n * p.L * 2 + (t - 64)];   // level-scaled points
  __syncthreads();

  for (int l = 0; l < p.nl; ++l) {
    const bool last = l == p.nl - 1;
    // opaque per iteration: otherwise every per-piece epilogue address derived from it (~160 of them) is hoisted out of the
    // layer loop and kept in registers for its whole body
    asm volatile("" : "+v"(i8));
    int ln = lane, tt = t;                                        // likewise for everything derived from the lane / thread id
    asm volatile("" : "+v"(ln), "+v"(tt));
    const long long crow = ((long long)n * p.T + p.step) * C;
    // the ring is (re)started per layer: carrying 128 VGPRs of in-flight weights over the loop edge made the register
    // allocator spill the whole ring at the loop header; one exposed round trip per layer is the cheaper price
    {
      const Blk b0 = blk(LW(w_qkv), C, 0, 767, 0);
#pragma unroll
      for (int c = 0; c < NQ; ++c) ld8(Q[c], b0, c);
    }
    // ================= A: q (folded projection + in_proj_q(query_pos)), k, v =================
    float4 x = *reinterpret_cast<const float4*>(sm + O_XIN + 4 * ln);
    const float4 qp = *reinterpret_cast<const float4*>(sm + O_QPOS + 4 * ln);
    {
      float acc[NQ][8];
      float bq[NQ];
#pragma unroll
      for (int c = 0; c < NQ; ++c) bq[c] = LW(b_qkv)[RP * c + wave + NW * i8];
      gemv256_acc(Q, x, acc, true, blk(LW(w_qin), C, 0, 255, 0));
      gemv256_acc(Q, qp, acc, false, blk(LW(w_qkv), C, 256, 767, 0));
      finish256(acc, bq, i8, [=](int j, float r) { sm[O_Q + j] = r; });
    }
    {
      float* const kc = LW(kc) + crow;
      float* const vc = LW(vc) + crow;
      gemv256(Q, x, LW(b_qkv), 256, 768, NQ, blk(LW(w_qkv), C, 512, 767, 0), i8, [=](int j, float r) { sm[O_KN + j - 256] = r; kc[j - 256] = r; });
      gemv256(Q, x, LW(b_qkv), 512, 768, NQ, blk(LW(w_o), C, 0, 255, 0), i8, [=](int j, float r) { sm[O_VN + j - 512] = r; vc[j - 512] = r; });
    }
    __syncthreads();
    // ================= B: self-attention over the cache rows 0..step-1 and this step's key =================
#ifndef LAB_NO_ATTN
    for (int h = wave; h < NH; h += NW)
      attn_head(sm + O_Q + h * HD, LW(kc) + (long long)n * p.T * C + h * HD, LW(vc) + (long long)n * p.T * C + h * HD, p.step, nullptr,
                sm + O_KN + h * HD, sm + O_VN + h * HD, sc, sm + O_ATT + h * HD, scale);
#endif
    __syncthreads();
    // ================= C: out_proj + residual -> p1; LN2 =================
    x = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * ln);
    gemv256(Q, x, LW(b_o), 0, 256, NQ, blk(LW(w_sq), C, 0, 255, 0), i8, [=](int j, float r) { sm[O_PRE + j] = r + sm[O_XIN + j]; });
    __syncthreads();
    x = ln256(sm + O_PRE, LW(g2), LW(be2));
    if (wave == 0) *reinterpret_cast<float4*>(sm + O_T + 4 * ln) = x;
    // ================= D: support cross-attention (host-checked: every layer has one) =================
    gemv256(Q, x, LW(b_sq), 0, 256, NQ, blk(LW(w_so), C, 0, 255, 0), i8, [=](int j, float r) { sm[O_Q + j] = r; });
    __syncthreads();
#ifndef LAB_NO_ATTN
    for (int h = wave; h < NH; h += NW)
      attn_head(sm + O_Q + h * HD, LW(supk) + (long long)n * p.P * C + h * HD, LW(supv) + (long long)n * p.P * C + h * HD, p.P,
                LW(supm) ? LW(supm) + (long long)n * p.P : nullptr, nullptr, nullptr, sc, sm + O_ATT + h * HD, scale);
#endif
    __syncthreads();
    x = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * ln);
    gemv256(Q, x, LW(b_so), 0, 256, NQ, blk(LW(w_off), C, 0, 383, 0), i8, [=](int j, float r) { sm[O_PRE + j] = r + sm[O_T + j]; });
    __syncthreads();
    x = ln256(sm + O_PRE, LW(gs), LW(bes));
    if (wave == 0) *reinterpret_cast<float4*>(sm + O_T2 + 4 * ln) = x;
    // ================= E: sampling offsets | attention logits of (tt + query_pos) =================
    {
      const float4 xq = make_float4(x.x + qp.x, x.y + qp.y, x.z + qp.z, x.w + qp.w);
      gemv256(Q, xq, LW(b_off), 0, 384, NQ, blk(LW(w_off), C, 256, 383, 0, NQ / 2), i8, [=](int j, float r) { sm[O_OFFW + j] = r; });
      gemv256(Q, xq, LW(b_off), 256, 384, NQ / 2, blk(LW(w_mo), C, 0, 255, 0), i8, [=](int j, float r) { sm[O_OFFW + j] = r; });
    }
    __syncthreads();
    // ================= F: deformable sampling of the cached value projection =================
    {
      const int LP = p.L * p.NP;
      float4* rec_w = reinterpret_cast<float4*>(sm + O_RECW);
      uint2* rec_i = reinterpret_cast<uint2*>(sm + O_RECI);
#ifndef LAB_NO_MSDA
      if (wave == 0) {
        const int h = ln >> 3, i = ln & 7;
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
        mx = fmaxf(mx, dppf<0xB1>(mx)); mx = fmaxf(mx, dppf<0x4E>(mx)); mx = fmaxf(mx, dppf<0x141>(mx));
        const float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx);
        float se = e0 + e1;
        se = sum8f(se);
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
        const int h = ln >> 3, i = ln & 7;
        const float* vb = LW(value) + (long long)n * p.S * C + h * HD + i * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int SPW = 16 / NW;                              // L * n_points == 16 (host-checked): samples per wave
        float4 g4[SPW][4];
#pragma unroll
        for (int q2 = 0; q2 < SPW; ++q2) {
          const uint2 ids = rec_i[h * 16 + SPW * wave + q2];
          g4[q2][0] = *reinterpret_cast<const float4*>(vb + (ids.x & 0xFFFFu) * (unsigned)C);
          g4[q2][1] = *reinterpret_cast<const float4*>(vb + (ids.x >> 16) * (unsigned)C);
          g4[q2][2] = *reinterpret_cast<const float4*>(vb + (ids.y & 0xFFFFu) * (unsigned)C);
          g4[q2][3] = *reinterpret_cast<const float4*>(vb + (ids.y >> 16) * (unsigned)C);
        }
#pragma unroll
        for (int q2 = 0; q2 < SPW; ++q2) {
          const float4 ww = rec_w[h * 16 + SPW * wave + q2];
          acc.x += ww.x * g4[q2][0].x + ww.y * g4[q2][1].x + ww.z * g4[q2][2].x + ww.w * g4[q2][3].x;
          acc.y += ww.x * g4[q2][0].y + ww.y * g4[q2][1].y + ww.z * g4[q2][2].y + ww.w * g4[q2][3].y;
          acc.z += ww.x * g4[q2][0].z + ww.y * g4[q2][1].z + ww.z * g4[q2][2].z + ww.w * g4[q2][3].z;
          acc.w += ww.x * g4[q2][0].w + ww.y * g4[q2][1].w + ww.z * g4[q2][2].w + ww.w * g4[q2][3].w;
        }
        *reinterpret_cast<float4*>(sm + O_PART + wave * C + h * HD + i * 4) = acc;
      }
#endif
      __syncthreads();
      if (tt < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) a += sm[O_PART + k * C + tt];
        sm[O_G + tt] = a;
      }
      __syncthreads();
    }
    // ================= G: output_proj + residual -> p3; LN1 =================
    x = *reinterpret_cast<const float4*>(sm + O_G + 4 * ln);
    gemv256(Q, x, LW(b_mo), 0, 256, NQ, blk(LW(w1), C, 0, FFN - 1, 0), i8, [=](int j, float r) { sm[O_PRE + j] = r + sm[O_T2 + j]; });
    __syncthreads();
    x = ln256(sm + O_PRE, LW(g1), LW(be1));
    if (wave == 0) *reinterpret_cast<float4*>(sm + O_T3 + 4 * ln) = x;
    // ================= H: feed-forward =================
    constexpr int nb1 = FFN / 256;
#pragma unroll
    for (int b = 0; b < nb1; ++b) {
      const Blk nxt = b + 1 < nb1 ? blk(LW(w1), C, 256 * (b + 1), FFN - 1, 0) : blk(LW(w2), FFN, 0, 255, 0);
      gemv256(Q, x, LW(b1), 256 * b, FFN, NQ, nxt, i8, [=](int j, float r) { sm[O_H + j] = fmaxf(r, 0.f); });
    }
    __syncthreads();
    {
      float acc[NQ][8];
      float b2v[NQ];
#pragma unroll
      for (int c = 0; c < NQ; ++c) b2v[c] = LW(b2)[RP * c + wave + NW * i8];
#pragma unroll
      for (int b = 0; b < nb1; ++b) {
        const float4 hx = *reinterpret_cast<const float4*>(sm + O_H + 256 * b + 4 * ln);
        const Blk nxt = b + 1 < nb1 ? blk(LW(w2), FFN, 0, 255, 256 * (b + 1)) : blk(LW(m1w), C, 0, 255, 0);
        gemv256_acc(Q, hx, acc, b == 0, nxt);
      }
      finish256(acc, b2v, i8, [=](int j, float r) { sm[O_PRE + j] = r + sm[O_T3 + j]; });
    }
    __syncthreads();
    // ================= I: tail -- LN3, class head, coords MLP, refinement, next query position embedding =================
    x = ln256(sm + O_PRE, LW(g3), LW(be3));
    if (wave == 0) {
      *reinterpret_cast<float4*>(sm + O_XIN + 4 * ln) = x;                           // next layer's input (and its residual)
      if (last) *reinterpret_cast<float4*>(p.out_hs + (long long)n * p.ld_hs + 4 * ln) = x;
    }
    if (last && wave < p.ncls) {
      const float4 wc = *reinterpret_cast<const float4*>(p.wc + (long long)wave * C + 4 * ln);
      const float c = wsum(fmaf(wc.x, x.x, fmaf(wc.y, x.y, fmaf(wc.z, x.z, wc.w * x.w))));
      if (ln == 0) p.out_logits[(long long)n * p.ld_logits + wave] = c + p.bc[wave];
    }
    gemv256(Q, x, LW(m1b), 0, 256, NQ, blk(LW(m2w), C, 0, 255, 0), i8, [=](int j, float r) { sm[O_G + j] = fmaxf(r, 0.f); });
    __syncthreads();
    x = *reinterpret_cast<const float4*>(sm + O_G + 4 * ln);
    gemv256(Q, x, LW(m2b), 0, 256, NQ, blk(p.wp, C, 0, 255, 0), i8, [=](int j, float r) { sm[O_ATT + j] = fmaxf(r, 0.f); });
    __syncthreads();
    if (wave < 2) {
      const float4 hh = *reinterpret_cast<const float4*>(sm + O_ATT + 4 * ln);
      const float4 w3 = *reinterpret_cast<const float4*>(LW(m3w) + (long long)wave * C + 4 * ln);
      const float dlt = wsum(fmaf(w3.x, hh.x, fmaf(w3.y, hh.y, fmaf(w3.z, hh.z, w3.w * hh.w))));
      if (ln == 0) {
        const float z = dlt + LW(m3b)[wave] + inv_sigmoid_d(small[wave]);
        const float r = 1.f / (1.f + expf(-z));
        small[4 + wave] = r;
        if (last) p.out_coords[(long long)n * p.ld_coords + wave] = r;
      }
    }
    __syncthreads();
    if (last) break;
    const float rx = small[4], ry = small[5];
    __syncthreads();                                              // everyone has read the refined point before it is republished
    if (tt < 2) small[tt] = tt ? ry : rx;                                                  // reference point of the next layer
    if (tt >= 64 && tt < 64 + 2 * p.L) {
      const int k = tt - 64;
      small[8 + k] = ((k & 1) ? ry : rx) * p.vr[(long long)n * p.L * 2 + k];
    }
    {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * ln + i, k = c & 127;
        const float a = ((c >> 7) ? ry : rx) * 6.283185307179586f / p.dim_t[k];
        e[i] = (k & 1) ? cosf(a) : sinf(a);
      }
      gemv256(Q, make_float4(e[0], e[1], e[2], e[3]), p.bp, 0, 256, NQ, blk(p.wp, C, 0, 255, 0, 0), i8,
              [=](int j, float r) { sm[O_PRE + j] = r; });
    }
    __syncthreads();
    {
      const float4 qn = ln256(sm + O_PRE, p.gp, p.bep);          // O_QPOS' old value was read at the top of this layer
      if (wave == 0) *reinterpret_cast<float4*>(sm + O_QPOS + 4 * ln) = qn;
    }
    __syncthreads();
  }
}

#undef LW
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
  CAPE_REQUIRE(d->ncls >= 1 && d->ncls <= NW, "cape_decode_step: ncls=%d (one wave per class, at most %d)", d->ncls, NW);
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
  const size_t lds = (size_t)(O_SC + NW * scw) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t max_lds = (size_t)(O_SC + NW * ((MAXKEYS + 4) & ~3)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return cape_set_error("cape_decode_step: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(decode_step_kernel, dim3(d->N), dim3(64 * NW), lds, as_stream(stream), p);
  CAPE_LAUNCH_CHECK("cape_decode_step");
  return 0;
}
