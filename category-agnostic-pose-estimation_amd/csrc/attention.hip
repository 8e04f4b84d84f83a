// attention.hip -- small dense attention core (head dim 32) forward (any length, keys tiled by 256) and backward (L <= 256).
//
// The CAPE decoder's sequences are tiny (L = 200 tokens, <= 68 support keypoints), so the whole K/V
// (or Q/dO) panel of one (image, head) lives in LDS (<= 64 KB) and each query (key) row is owned by a
// group of 4 lanes x 8 channels: dot products finish with two xor-shuffles, LDS reads of a key row
// are wave-wide broadcasts (4 distinct addresses), no atomics anywhere:
//   fwd      : thread group per query row, online softmax, saves LSE
//   bwd dQ   : thread group per query row (recomputes P from LSE)
//   bwd dK/dV: thread group per key row, loops over the queries
#include "common.h"

namespace {

constexpr int HD = 32;       // head dim
constexpr int MAXL = 256;    // max staged rows
constexpr int ROWS = 64;     // rows per block (x 4 lanes)

struct AttnP {
  long long ldq, ldk, ldv, ldo;
  long long bsq, bsk, bsv, bso;   // batch strides (elements)
  int N, H, Lq, Lk;
  float scale;
  int mask_mode, causal_offset;
  const uint8_t* kpm;
  uint32_t thresh; float inv_keep;
  const uint64_t* rng_state; uint32_t rng_stream;
};

// sum over the 4 lanes of a quad with two DPP quad permutes (xor 1: [1,0,3,2] = 0xB1, xor 2: [2,3,0,1] = 0x4E); the
// __shfl_xor form compiled to ds_bpermute_b32, an LDS-pipe round trip inside the per-key dependency chain
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  return v;
}

// cooperative load of `rows` x 32 floats (row stride ld) into LDS [rows][32]
__device__ __forceinline__ void stage_rows(float* dst, const float* src, long long ld, int rows) {
  for (int idx = threadIdx.x; idx < rows * 8; idx += blockDim.x) {
    const int r = idx >> 3, c = (idx & 7) * 4;
    *reinterpret_cast<float4*>(dst + r * HD + c) = *reinterpret_cast<const float4*>(src + (long long)r * ld + c);
  }
}

__device__ __forceinline__ float dot8(const float* a, const float4 b0, const float4 b1) {
  return a[0] * b0.x + a[1] * b0.y + a[2] * b0.z + a[3] * b0.w + a[4] * b1.x + a[5] * b1.y + a[6] * b1.z + a[7] * b1.w;
}

__global__ void __launch_bounds__(256) attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, float* __restrict__ O,
                                                        float* __restrict__ lse, const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // keys are staged in chunks of <= MAXL rows (one chunk for the CAPE decoder's L <= 200; image-token panels of the
  // bidirectional attention blocks take several), the online softmax state carries across chunks
  const int KC = min(p.Lk, MAXL);
  float* Ks = smem;
  float* Vs = smem + KC * HD;
  const int n = blockIdx.z, h = blockIdx.y;
  const int sub = threadIdx.x & 3;
  const int i = blockIdx.x * ROWS + (threadIdx.x >> 2);
  const bool live = i < p.Lq;
  float q[8], acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { q[d] = 0.f; acc[d] = 0.f; }
  if (live) {
    const float* qp = Q + (long long)n * p.bsq + (long long)i * p.ldq + h * HD + sub * 8;
    const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
    q[0] = a.x * p.scale; q[1] = a.y * p.scale; q[2] = a.z * p.scale; q[3] = a.w * p.scale;
    q[4] = b.x * p.scale; q[5] = b.y * p.scale; q[6] = b.z * p.scale; q[7] = b.w * p.scale;
  }
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  float m = -INFINITY, l = 0.f;
  // wave-uniform loop bound (all lanes must take part in the shuffles)
  int jend = p.Lk;
  if (p.mask_mode == 1) {
    const int last_row = min(p.Lq - 1, blockIdx.x * ROWS + ((threadIdx.x | 63) >> 2));
    jend = min(p.Lk, last_row + p.causal_offset + 1);
  }
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const uint64_t rbase = (((uint64_t)n * p.H + h) * p.Lq + (uint64_t)(live ? i : 0)) * p.Lk;
  for (int k0 = 0; k0 < p.Lk; k0 += MAXL) {
    const int kc = min(MAXL, p.Lk - k0);
    if (k0) __syncthreads();                   // everyone is done with the previous chunk
    stage_rows(Ks, K + (long long)n * p.bsk + (long long)k0 * p.ldk + h * HD, p.ldk, kc);
    stage_rows(Vs, V + (long long)n * p.bsv + (long long)k0 * p.ldv + h * HD, p.ldv, kc);
    __syncthreads();
    const int jhi = min(kc, jend - k0);        // wave-uniform
    for (int jj = 0; jj < jhi; ++jj) {
      const int j = k0 + jj;
      const float4 k0v = *reinterpret_cast<const float4*>(Ks + jj * HD + sub * 8);
      const float4 k1v = *reinterpret_cast<const float4*>(Ks + jj * HD + sub * 8 + 4);
      float s = quad_sum(dot8(q, k0v, k1v));
      bool masked = !live;
      if (p.mask_mode == 1) masked = masked || (j > i + p.causal_offset);
      if (kp) masked = masked || (kp[j] != 0);
      if (masked) continue;                    // after the shuffle: safe
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn);      // m = -inf on the first unmasked key -> 0
      const float pe = __expf(s - mn);
      l = l * alpha + pe;
      float pd = pe;
      if (p.thresh) pd = cape_keep(seed, step, p.rng_stream, rbase + j, p.thresh) ? pe * p.inv_keep : 0.f;
      const float4 v0 = *reinterpret_cast<const float4*>(Vs + jj * HD + sub * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Vs + jj * HD + sub * 8 + 4);
      acc[0] = acc[0] * alpha + pd * v0.x; acc[1] = acc[1] * alpha + pd * v0.y;
      acc[2] = acc[2] * alpha + pd * v0.z; acc[3] = acc[3] * alpha + pd * v0.w;
      acc[4] = acc[4] * alpha + pd * v1.x; acc[5] = acc[5] * alpha + pd * v1.y;
      acc[6] = acc[6] * alpha + pd * v1.z; acc[7] = acc[7] * alpha + pd * v1.w;
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.f / l;                 // l == 0 (fully masked row) -> NaN like torch
    float* op = O + (long long)n * p.bso + (long long)i * p.ldo + h * HD + sub * 8;
    *reinterpret_cast<float4*>(op) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<float4*>(op + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
    if (sub == 0) lse[((long long)n * p.H + h) * p.Lq + i] = m + __logf(l);
  }
}

// Single-query form (the cached decode step: Lq == 1, no dropout): one wave per (image, head).  The general kernel
// would keep one 4-lane group busy per block walking the keys serially behind a 256-thread staging pass (19 us per call);
// here the 64 lanes split the keys for q.k (a key row is one 128-byte read per lane), the softmax is two wave
// reductions, and P.V runs with lane = channel (coalesced V rows), the two half-waves taking alternate keys.
constexpr int DEC_MAXL = 1024;

__global__ void __launch_bounds__(64) attn_decode_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, float* __restrict__ O,
                                                         float* __restrict__ lse, const AttnP p) {
  __shared__ float sc[DEC_MAXL];
  const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const float* qp = Q + (long long)n * p.bsq + h * HD;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 a = *reinterpret_cast<const float4*>(qp + d);
    q[d] = a.x * p.scale; q[d + 1] = a.y * p.scale; q[d + 2] = a.z * p.scale; q[d + 3] = a.w * p.scale;
  }
  const int jmax = p.mask_mode == 1 ? min(p.Lk, p.causal_offset + 1) : p.Lk;
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const float* kb = K + (long long)n * p.bsk + h * HD;
  float m = -INFINITY;
  for (int j = lane; j < jmax; j += 64) {
    const float* kr = kb + (long long)j * p.ldk;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 b = *reinterpret_cast<const float4*>(kr + d);
      s += q[d] * b.x + q[d + 1] * b.y + q[d + 2] * b.z + q[d + 3] * b.w;
    }
    if (kp && kp[j]) s = -INFINITY;
    sc[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < jmax; j += 64) {
    const float e = __expf(sc[j] - m);             // a fully masked row: m = -inf -> NaN like torch
    sc[j] = e;
    l += e;
  }
  l = wave_sum(l);
  __syncthreads();                                 // one wave: orders the LDS writes above before the reads below
  const int c = lane & 31, half = lane >> 5;
  const float* vb = V + (long long)n * p.bsv + h * HD + c;
  // four independent partial sums: the V rows of four keys are in flight together (a single chain serialises the loads)
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int j = half;
  for (; j + 6 < jmax; j += 8) {
    const float v0 = vb[(long long)j * p.ldv], v1 = vb[(long long)(j + 2) * p.ldv];
    const float v2 = vb[(long long)(j + 4) * p.ldv], v3 = vb[(long long)(j + 6) * p.ldv];
    acc0 = fmaf(sc[j], v0, acc0); acc1 = fmaf(sc[j + 2], v1, acc1); acc2 = fmaf(sc[j + 4], v2, acc2); acc3 = fmaf(sc[j + 6], v3, acc3);
  }
  for (; j < jmax; j += 2) acc0 = fmaf(sc[j], vb[(long long)j * p.ldv], acc0);
  float acc = (acc0 + acc1) + (acc2 + acc3);
  acc += __shfl_xor(acc, 32, 64);
  if (lane < 32) O[(long long)n * p.bso + h * HD + c] = acc / l;
  if (lane == 0) lse[(long long)n * p.H + h] = m + __logf(l);
}

// dQ: thread group per query row
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const float* __restrict__ dO, const float* __restrict__ Q,
                                                           const float* __restrict__ K, const float* __restrict__ V,
                                                           const float* __restrict__ O, const float* __restrict__ lse,
                                                           float* __restrict__ dQ, const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // keys / values are staged in panels of <= MAXL rows (one panel for every sequence of the CAPE step; the bidirectional
  // blocks' 280-patch panels take two): the row's logsumexp is known, so the panels simply accumulate
  const int KC = min(p.Lk, MAXL);
  float* Ks = smem;
  float* Vs = smem + KC * HD;
  const int n = blockIdx.z, h = blockIdx.y;
  const int sub = threadIdx.x & 3;
  const int i = blockIdx.x * ROWS + (threadIdx.x >> 2);
  const bool live = i < p.Lq;
  float q[8], go[8], acc[8];
  float D = 0.f, L = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) { q[d] = 0.f; go[d] = 0.f; acc[d] = 0.f; }
  if (live) {
    const float* qp = Q + (long long)n * p.bsq + (long long)i * p.ldq + h * HD + sub * 8;
    const float* gp = dO + (long long)n * p.bso + (long long)i * p.ldo + h * HD + sub * 8;
    const float* op = O + (long long)n * p.bso + (long long)i * p.ldo + h * HD + sub * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) { q[d] = qp[d] * p.scale; go[d] = gp[d]; D += gp[d] * op[d]; }
    L = lse[((long long)n * p.H + h) * p.Lq + i];
  }
  D = quad_sum(D);
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  int jend = p.Lk;
  if (p.mask_mode == 1) {
    const int last_row = min(p.Lq - 1, blockIdx.x * ROWS + ((threadIdx.x | 63) >> 2));
    jend = min(p.Lk, last_row + p.causal_offset + 1);
  }
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const uint64_t rbase = (((uint64_t)n * p.H + h) * p.Lq + (uint64_t)(live ? i : 0)) * p.Lk;
  for (int kb = 0; kb < p.Lk; kb += MAXL) {
  const int kc = min(MAXL, p.Lk - kb);
  if (kb) __syncthreads();                     // everyone is done with the previous panel
  stage_rows(Ks, K + (long long)n * p.bsk + (long long)kb * p.ldk + h * HD, p.ldk, kc);
  stage_rows(Vs, V + (long long)n * p.bsv + (long long)kb * p.ldv + h * HD, p.ldv, kc);
  __syncthreads();
  const int jhi = min(kb + kc, jend);          // wave-uniform
  for (int j = kb; j < jhi; ++j) {
    const int jj = j - kb;
    const float4 k0 = *reinterpret_cast<const float4*>(Ks + jj * HD + sub * 8);
    const float4 k1 = *reinterpret_cast<const float4*>(Ks + jj * HD + sub * 8 + 4);
    const float4 v0 = *reinterpret_cast<const float4*>(Vs + jj * HD + sub * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(Vs + jj * HD + sub * 8 + 4);
    const float s = quad_sum(dot8(q, k0, k1));
    float dp = quad_sum(dot8(go, v0, v1));
    bool masked = !live;
    if (p.mask_mode == 1) masked = masked || (j > i + p.causal_offset);
    if (kp) masked = masked || (kp[j] != 0);
    if (masked) continue;
    const float pe = __expf(s - L);
    if (p.thresh) dp = cape_keep(seed, step, p.rng_stream, rbase + j, p.thresh) ? dp * p.inv_keep : 0.f;
    const float ds = pe * (dp - D) * p.scale;
    acc[0] += ds * k0.x; acc[1] += ds * k0.y; acc[2] += ds * k0.z; acc[3] += ds * k0.w;
    acc[4] += ds * k1.x; acc[5] += ds * k1.y; acc[6] += ds * k1.z; acc[7] += ds * k1.w;
  }
  }
  if (live) {
    float* dp_ = dQ + (long long)n * p.bsq + (long long)i * p.ldq + h * HD + sub * 8;
    *reinterpret_cast<float4*>(dp_) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dp_ + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// dK, dV: thread group per key row
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const float* __restrict__ dO, const float* __restrict__ Q,
                                                            const float* __restrict__ K, const float* __restrict__ V,
                                                            const float* __restrict__ O, const float* __restrict__ lse,
                                                            float* __restrict__ dK, float* __restrict__ dV, const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int QC = min(p.Lq, MAXL);        // queries are staged in panels of <= MAXL rows
  float* Qs = smem;                       // [QC][32]
  float* Gs = smem + QC * HD;             // dO [QC][32]
  float* Ds = smem + 2 * QC * HD;         // D_i
  float* Ls = Ds + QC;                    // lse_i
  const int n = blockIdx.z, h = blockIdx.y;
  const int sub = threadIdx.x & 3;
  // few keys (the 17 support keypoints): the 4-lane key groups would fill a quarter of the block and walk all Lq queries alone;
  // the block is cut into `parts` partitions of 256 / parts threads that take every parts-th query and meet in LDS at the end
  const int parts = p.Lk <= 16 ? 4 : (p.Lk <= 32 ? 2 : 1);
  const int tpp = 256 / parts, part = threadIdx.x / tpp, tl = threadIdx.x - part * tpp;
  const int j = blockIdx.x * (tpp >> 2) + (tl >> 2);
  const bool live = j < p.Lk;
  float k[8], v[8], ak[8], av[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { k[d] = 0.f; v[d] = 0.f; ak[d] = 0.f; av[d] = 0.f; }
  bool keymasked = false;
  if (live) {
    const float* kp_ = K + (long long)n * p.bsk + (long long)j * p.ldk + h * HD + sub * 8;
    const float* vp_ = V + (long long)n * p.bsv + (long long)j * p.ldv + h * HD + sub * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) { k[d] = kp_[d] * p.scale; v[d] = vp_[d]; }
    if (p.mask_mode == 2) keymasked = p.kpm[(long long)n * p.Lk + j] != 0;
  }
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  // causal: key j is seen by queries i >= j - offset ; wave-uniform start = min over the wave's keys
  int ibeg = 0;
  if (p.mask_mode == 1) ibeg = max(0, blockIdx.x * (tpp >> 2) + ((tl & ~63) >> 2) - p.causal_offset);
  for (int qb = 0; qb < p.Lq; qb += MAXL) {
  const int qc = min(MAXL, p.Lq - qb);
  if (qb) __syncthreads();                     // everyone is done with the previous panel
  stage_rows(Qs, Q + (long long)n * p.bsq + (long long)qb * p.ldq + h * HD, p.ldq, qc);
  stage_rows(Gs, dO + (long long)n * p.bso + (long long)qb * p.ldo + h * HD, p.ldo, qc);
  for (int r = threadIdx.x; r < qc; r += blockDim.x) {
    const float* op = O + (long long)n * p.bso + (long long)(qb + r) * p.ldo + h * HD;
    const float* gp = dO + (long long)n * p.bso + (long long)(qb + r) * p.ldo + h * HD;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) d += op[c] * gp[c];
    Ds[r] = d;
    Ls[r] = lse[((long long)n * p.H + h) * p.Lq + qb + r];
  }
  __syncthreads();
  for (int i = max(ibeg, qb) + part; i < qb + qc; i += parts) {      // (any split of a panel's queries over the partitions will do)
    const int ii = i - qb;
    const float4 q0 = *reinterpret_cast<const float4*>(Qs + ii * HD + sub * 8);
    const float4 q1 = *reinterpret_cast<const float4*>(Qs + ii * HD + sub * 8 + 4);
    const float4 g0 = *reinterpret_cast<const float4*>(Gs + ii * HD + sub * 8);
    const float4 g1 = *reinterpret_cast<const float4*>(Gs + ii * HD + sub * 8 + 4);
    const float s = quad_sum(dot8(k, q0, q1));
    float dp = quad_sum(dot8(v, g0, g1));
    bool masked = !live || keymasked;
    if (p.mask_mode == 1) masked = masked || (j > i + p.causal_offset);
    if (masked) continue;
    const float pe = __expf(s - Ls[ii]);
    float pd = pe;
    if (p.thresh) {
      const bool keep = cape_keep(seed, step, p.rng_stream, (((uint64_t)n * p.H + h) * p.Lq + i) * p.Lk + j, p.thresh);
      pd = keep ? pe * p.inv_keep : 0.f;
      dp = keep ? dp * p.inv_keep : 0.f;
    }
    av[0] += pd * g0.x; av[1] += pd * g0.y; av[2] += pd * g0.z; av[3] += pd * g0.w;
    av[4] += pd * g1.x; av[5] += pd * g1.y; av[6] += pd * g1.z; av[7] += pd * g1.w;
    const float ds = pe * (dp - Ds[ii]) * p.scale;
    ak[0] += ds * q0.x; ak[1] += ds * q0.y; ak[2] += ds * q0.z; ak[3] += ds * q0.w;
    ak[4] += ds * q1.x; ak[5] += ds * q1.y; ak[6] += ds * q1.z; ak[7] += ds * q1.w;
  }
  }
  if (parts > 1) {                                               // block-uniform
    __syncthreads();                                             // every partition is done with Qs / Gs
    float* red = smem;                                           // [part][tl][16]
    if (part > 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) { red[(threadIdx.x) * 16 + d] = ak[d]; red[(threadIdx.x) * 16 + 8 + d] = av[d]; }
    }
    __syncthreads();
    if (part == 0) {
      for (int q = 1; q < parts; ++q) {
#pragma unroll
        for (int d = 0; d < 8; ++d) { ak[d] += red[(q * tpp + tl) * 16 + d]; av[d] += red[(q * tpp + tl) * 16 + 8 + d]; }
      }
    }
  }
  if (live && part == 0) {
    float* dk = dK + (long long)n * p.bsk + (long long)j * p.ldk + h * HD + sub * 8;
    float* dv = dV + (long long)n * p.bsv + (long long)j * p.ldv + h * HD + sub * 8;
    *reinterpret_cast<float4*>(dk) = make_float4(ak[0], ak[1], ak[2], ak[3]);
    *reinterpret_cast<float4*>(dk + 4) = make_float4(ak[4], ak[5], ak[6], ak[7]);
    *reinterpret_cast<float4*>(dv) = make_float4(av[0], av[1], av[2], av[3]);
    *reinterpret_cast<float4*>(dv + 4) = make_float4(av[4], av[5], av[6], av[7]);
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core form of the attention core for long sequences (the decoder's 200 x 200 causal self-attention): the two
// contractions Q K^T and P V (and the four of the backward) run as batched launches of the GEMM family (one product per
// (image, head), bf16x3 or exact fp32 MFMA), and only the row softmax stays here: one wave per score row.
//   forward : P = softmax(scale * S + mask) ; Pd = dropout(P)            (S, P, Pd: (N, H, Lq, Lk) fp32)
//   backward: dS = scale * P o (dPd_eff - sum_j dPd_eff P)  with dPd_eff = dropout-mask o dPd / keep      (in place)
// The dropout decision uses the same counter-based index ((n*H + h)*Lq + i)*Lk + j as the fused kernels above.
// ---------------------------------------------------------------------------------------------
// (rows of up to 256 scores live in registers -- four per lane -- between the passes: one read of S / dS, one hash per element)
__global__ void __launch_bounds__(256) attn_softmax_fwd_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                               float* __restrict__ Pd, const AttnP p) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // (n*H + h)*Lq + i
  const long long rows = (long long)p.N * p.H * p.Lq;
  if (row >= rows) return;                                                   // wave-uniform
  const int lane = threadIdx.x & 63;
  const int i = (int)(row % p.Lq);
  const int n = (int)(row / ((long long)p.H * p.Lq));
  const float* s = S + row * p.Lk;
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const int jend = p.mask_mode == 1 ? min(p.Lk, i + p.causal_offset + 1) : p.Lk;
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  if (p.Lk <= 256) {
    float v[4];
    bool live[4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      live[k] = j < jend && !(kp && kp[j]);
      v[k] = live[k] ? s[j] * p.scale : -INFINITY;
      m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = live[k] ? __expf(v[k] - m) : 0.f; l += v[k]; }
    l = wave_sum(l);
    const float inv = 1.f / l;                                               // fully masked row: 0/0 -> NaN like torch
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      if (j < p.Lk) {
        const float pv = live[k] ? v[k] * inv : (l == 0.f ? NAN : 0.f);
        P[row * p.Lk + j] = pv;
        if (Pd) Pd[row * p.Lk + j] = cape_keep(seed, step, p.rng_stream, (uint64_t)row * p.Lk + j, p.thresh) ? pv * p.inv_keep : 0.f;
      }
    }
    return;
  }
  float m = -INFINITY;
  for (int j = lane; j < jend; j += 64)
    if (!(kp && kp[j])) m = fmaxf(m, s[j] * p.scale);
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < jend; j += 64)
    if (!(kp && kp[j])) l += __expf(s[j] * p.scale - m);
  l = wave_sum(l);
  const float inv = 1.f / l;
  for (int j = lane; j < p.Lk; j += 64) {
    const bool live = j < jend && !(kp && kp[j]);
    const float pv = live ? __expf(s[j] * p.scale - m) * inv : (l == 0.f ? NAN : 0.f);
    P[row * p.Lk + j] = pv;
    if (Pd) Pd[row * p.Lk + j] = cape_keep(seed, step, p.rng_stream, (uint64_t)row * p.Lk + j, p.thresh) ? pv * p.inv_keep : 0.f;
  }
}

__global__ void __launch_bounds__(256) attn_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dS, const AttnP p) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long rows = (long long)p.N * p.H * p.Lq;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  const float* pr = P + row * p.Lk;
  float* d = dS + row * p.Lk;
  if (p.Lk <= 256) {
    float g[4], pv[4];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      g[k] = 0.f; pv[k] = 0.f;
      if (j < p.Lk) {
        g[k] = d[j]; pv[k] = pr[j];
        if (p.thresh) g[k] = cape_keep(seed, step, p.rng_stream, (uint64_t)row * p.Lk + j, p.thresh) ? g[k] * p.inv_keep : 0.f;
      }
      t += g[k] * pv[k];
    }
    t = wave_sum(t);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      if (j < p.Lk) d[j] = p.scale * pv[k] * (g[k] - t);
    }
    return;
  }
  float t = 0.f;
  for (int j = lane; j < p.Lk; j += 64) {
    float g = d[j];
    if (p.thresh) g = cape_keep(seed, step, p.rng_stream, (uint64_t)row * p.Lk + j, p.thresh) ? g * p.inv_keep : 0.f;
    t += g * pr[j];
  }
  t = wave_sum(t);
  for (int j = lane; j < p.Lk; j += 64) {
    float g = d[j];
    if (p.thresh) g = cape_keep(seed, step, p.rng_stream, (uint64_t)row * p.Lk + j, p.thresh) ? g * p.inv_keep : 0.f;
    d[j] = p.scale * pr[j] * (g - t);
  }
}

int fill(AttnP& p, long long ldq, long long ldk, long long ldv, long long ldo, const long long* bs, int N, int H, int Lq, int Lk, float scale,
         int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
         uint32_t rng_stream, bool fwd = false) {
  // forward and backward stage their panels MAXL rows at a time and take any length
  CAPE_REQUIRE(Lq >= 1 && Lk >= 1, "cape_attn: Lq=%d Lk=%d must be positive", Lq, Lk);
  CAPE_REQUIRE((ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (ldo % 4) == 0, "cape_attn: row strides must be multiples of 4");
  CAPE_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "cape_attn: bad mask_mode %d", mask_mode);
  CAPE_REQUIRE(mask_mode != 2 || kpm, "cape_attn: key padding mask missing");
  CAPE_REQUIRE(dropout_p == 0.f || (rng_state && dropout_p < 1.f), "cape_attn: dropout needs rng_state");
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.bsq = bs[0]; p.bsk = bs[1]; p.bsv = bs[2]; p.bso = bs[3];
  CAPE_REQUIRE((bs[0] % 4) == 0 && (bs[1] % 4) == 0 && (bs[2] % 4) == 0 && (bs[3] % 4) == 0, "cape_attn: batch strides must be multiples of 4");
  p.N = N; p.H = H; p.Lq = Lq; p.Lk = Lk; p.scale = scale;
  p.mask_mode = mask_mode; p.causal_offset = causal_offset; p.kpm = kpm;
  p.thresh = dropout_p > 0.f ? cape_drop_threshold(dropout_p) : 0u;
  p.inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  p.rng_state = rng_state; p.rng_stream = rng_stream;
  return 0;
}

int raise_lds_limit() {
  static bool done = false;
  if (done) return 0;
  const int lim = 96 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  if (e != hipSuccess) return cape_set_error("cape_attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
  done = true;
  return 0;
}

}  // namespace

extern "C" int cape_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, long long ldq,
                             long long ldk, long long ldv, long long ldo, long long bsq, long long bsk, long long bsv,
                             long long bso, int N, int H, int Lq, int Lk, float scale,
                             int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p,
                             const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(Q && K && V && O && lse, "cape_attn_fwd: null pointer");
  if (N <= 0 || H <= 0) return 0;
  AttnP p;
  const long long bs[4] = {bsq, bsk, bsv, bso};
  if (fill(p, ldq, ldk, ldv, ldo, bs, N, H, Lq, Lk, scale, mask_mode, causal_offset, kpm, dropout_p, rng_state, rng_stream, true)) return 1;
  if (Lq == 1 && dropout_p == 0.f && Lk <= DEC_MAXL && H <= 65535 && N <= 65535) {
    hipLaunchKernelGGL(attn_decode_kernel, dim3(H, N), dim3(64), 0, as_stream(stream), Q, K, V, O, lse, p);
    CAPE_LAUNCH_CHECK("cape_attn_fwd(decode)");
    return 0;
  }
  if (raise_lds_limit()) return 1;
  CAPE_REQUIRE((Lq + ROWS - 1) / ROWS <= 65535 * 32, "cape_attn_fwd: Lq too large");
  const size_t sh = (size_t)2 * (Lk < MAXL ? Lk : MAXL) * HD * sizeof(float);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((Lq + ROWS - 1) / ROWS, H, N), dim3(256), sh, as_stream(stream), Q, K, V, O, lse, p);
  CAPE_LAUNCH_CHECK("cape_attn_fwd");
  return 0;
}

extern "C" int cape_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* O,
                             const float* lse, float* dQ, float* dK, float* dV, long long ldq, long long ldk,
                             long long ldv, long long ldo, long long bsq, long long bsk, long long bsv, long long bso,
                             int N, int H, int Lq, int Lk, float scale, int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
                             uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(dO && Q && K && V && O && lse && dQ && dK && dV, "cape_attn_bwd: null pointer");
  if (N <= 0 || H <= 0) return 0;
  AttnP p;
  const long long bs[4] = {bsq, bsk, bsv, bso};
  if (fill(p, ldq, ldk, ldv, ldo, bs, N, H, Lq, Lk, scale, mask_mode, causal_offset, kpm, dropout_p, rng_state, rng_stream)) return 1;
  if (raise_lds_limit()) return 1;
  const int KC = Lk < MAXL ? Lk : MAXL, QC = Lq < MAXL ? Lq : MAXL;
  const size_t sh1 = (size_t)2 * KC * HD * sizeof(float);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((Lq + ROWS - 1) / ROWS, H, N), dim3(256), sh1, as_stream(stream), dO, Q, K, V,
                     O, lse, dQ, p);
  size_t sh2 = ((size_t)2 * QC * HD + 2 * QC) * sizeof(float);
  if (Lk <= 32 && sh2 < 256 * 16 * sizeof(float)) sh2 = 256 * 16 * sizeof(float);      // partition reduction scratch of the few-key form
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((Lk + ROWS - 1) / ROWS, H, N), dim3(256), sh2, as_stream(stream), dO, Q, K, V,
                     O, lse, dK, dV, p);
  CAPE_LAUNCH_CHECK("cape_attn_bwd");
  return 0;
}

extern "C" int cape_attn_softmax_fwd(const float* S, float* P, float* Pd, int N, int H, int Lq, int Lk, float scale, int mask_mode,
                                     int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
                                     uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(S && P, "cape_attn_softmax_fwd: null pointer");
  CAPE_REQUIRE((dropout_p > 0.f) == (Pd != nullptr), "cape_attn_softmax_fwd: Pd goes with dropout");
  if (N <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  AttnP p;
  const long long bs[4] = {0, 0, 0, 0};
  if (fill(p, 4, 4, 4, 4, bs, N, H, Lq, Lk, scale, mask_mode, causal_offset, kpm, dropout_p, rng_state, rng_stream, true)) return 1;
  const long long rows = (long long)N * H * Lq;
  CAPE_REQUIRE((rows + 3) / 4 < (1ll << 31), "cape_attn_softmax_fwd: too many rows");
  hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), S, P, Pd, p);
  CAPE_LAUNCH_CHECK("cape_attn_softmax_fwd");
  return 0;
}

extern "C" int cape_attn_softmax_bwd(const float* P, float* dS, int N, int H, int Lq, int Lk, float scale, float dropout_p,
                                     const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(P && dS, "cape_attn_softmax_bwd: null pointer");
  if (N <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  AttnP p;
  const long long bs[4] = {0, 0, 0, 0};
  if (fill(p, 4, 4, 4, 4, bs, N, H, Lq, Lk, scale, 0, 0, nullptr, dropout_p, rng_state, rng_stream, true)) return 1;
  const long long rows = (long long)N * H * Lq;
  CAPE_REQUIRE((rows + 3) / 4 < (1ll << 31), "cape_attn_softmax_bwd: too many rows");
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), P, dS, p);
  CAPE_LAUNCH_CHECK("cape_attn_softmax_bwd");
  return 0;
}
