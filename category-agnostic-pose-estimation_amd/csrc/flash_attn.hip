// flash_attn.hip -- fused attention core on the matrix cores for the decoder's L = 200 self-attention (head dim 32, rows of up
// to 224 keys): Q K^T, mask, softmax, dropout and P V in one kernel, backward in two (dQ; dK | dV) that recompute P from the
// saved log-sum-exp.  Replaces, per decoder layer, 2 + 4 batched GEMM launches around two softmax passes with the (N, H, L, L)
// score / probability / dropped-probability tensors (41 MB each) going through HBM (nn.MultiheadAttention core,
// deformable_transformer_v2.py:323-341).
//
// Layout of the work (MI355X): one block per (image, head), one wave per 32-row block (7 waves for L = 200).  Everything is
// arranged so that no score ever leaves the registers of the wave that computed it:
//   * forward and dQ work on TRANSPOSED tiles  T = K_j Q_i^T  (32 keys x 32 queries): in the 32x32 MFMA accumulator layout a
//     lane owns one column, i.e. one query, and its 16 registers are 16 keys of the tile -- row max / row sum of the softmax
//     are in-register loops plus ONE exchange between the two half waves (v_permlane32_swap), the log-sum-exp / D of the
//     query are lane-local scalars, and the tile is, as it stands, the B operand of the next product that sums over keys
//     (O^T = V^T P^T, dQ^T = K^T dS^T): registers 8s .. 8s+7 of the accumulator are the fragment of k-step s;
//   * dK | dV work on the plain tiles  S = Q_i K_j^T  (queries x keys) for the same reason: dV^T = dO^T P and dK^T = Q^T dS sum
//     over the tile's rows;
//   * the operand that such a product needs "the other way round" ([channel][key] instead of [key][channel]) is the only
//     thing staged through LDS: V^T (forward), K^T (dQ), Q^T and dO^T (dK | dV), as bf16 (hi, lo) planes with a 456-byte row
//     stride (conflict-free 8-byte fragment reads); all other fragments are 32-byte runs of the tensors in global memory.
// Arithmetic: the bf16x3 split of the GEMM family (x = hi + lo, three MFMAs per product, fp32 accumulate); the exact-fp32
// precision mode keeps the batched-GEMM form (hip/ops.attn_mm_*).  Dropout: the counter RNG of common.h with the element
// index ((n H + h) Lq + i) Lk + j of the other attention kernels, so forward and backward regenerate the same mask.
#include "gemm_common.h"

namespace {

constexpr int FA_MAXB = 7;                 // 32-row blocks per sequence (L <= 224)
constexpr int FA_LD = 228;                 // bf16 elements per transposed LDS row (456 bytes)
constexpr int FA_PLANE = 32 * FA_LD;       // one plane: 32 channels

struct FlashP {
  long long ldq, ldk, ldv, ldo;            // row strides (elements) of Q / K / V / O (dQ, dK, dV share ldq, ldk, ldv; dO shares ldo)
  long long bsq, bsk, bsv, bso;            // image strides
  int N, H, Lq, Lk;
  float scale;
  int mask_mode, causal_offset;            // 0 none, 1 causal (key <= query + offset), 2 key padding mask
  const uint8_t* kpm;
  uint32_t thresh; float inv_keep;
  const uint64_t* rng_state; uint32_t rng_stream;
};

__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float half_max(float v) { float a = v, b = v; swap32(a, b); return fmaxf(a, b); }   // over lanes l, l ^ 32
__device__ __forceinline__ float half_sum(float v) { float a = v, b = v; swap32(a, b); return a + b; }

// 8 consecutive floats -> bf16x8 (hi) and bf16x8 (lo = x - hi)
__device__ __forceinline__ void split8(float4 a, float4 b, bf16x8& hi, bf16x8& lo) {
  unsigned h[4], l[4];
  split2(a.x, a.y, h[0], l[0]); split2(a.z, a.w, h[1], l[1]);
  split2(b.x, b.y, h[2], l[2]); split2(b.z, b.w, h[3], l[3]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// fragment of a row-major tensor for one 32x32x16 step: lane (r, h) takes X[row0 + r][c0 + 8 h .. + 8) (rows clamped to nrows - 1)
struct Frag2 { bf16x8 hi[2], lo[2]; };
__device__ __forceinline__ Frag2 load_frag(const float* base, long long ld, int row0, int nrows, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* p = base + (long long)min(row0 + r, nrows - 1) * ld + 8 * h;
  Frag2 f;
#pragma unroll
  for (int s = 0; s < 2; ++s) split8(*reinterpret_cast<const float4*>(p + 16 * s), *reinterpret_cast<const float4*>(p + 16 * s + 4), f.hi[s], f.lo[s]);
  return f;
}

__device__ __forceinline__ f32x16 mma3(const bf16x8& ahi, const bf16x8& alo, const bf16x8& bhi, const bf16x8& blo, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 tile_product(const Frag2& a, const Frag2& b) {      // A (32 x 32 channels) . B^T over the 32 channels
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = mma3(a.hi[0], a.lo[0], b.hi[0], b.lo[0], acc);
  return mma3(a.hi[1], a.lo[1], b.hi[1], b.lo[1], acc);
}

// rows of X (nrows x 32 channels of one head, row stride ld) -> transposed bf16 planes Xt[plane][channel][row] in LDS, rows beyond
// nrows zero.  448 threads = 112 row pairs x 4 channel groups: one 32-bit word (rows 2 kp, 2 kp + 1) per channel and plane.
__device__ __forceinline__ void stage_transposed(unsigned short* dst, const float* src, long long ld, int nrows) {
  const int t = threadIdx.x;
  const int kp = t >> 2, cg = t & 3;
  float4 a0 = zero4(), a1 = zero4(), b0 = zero4(), b1 = zero4();
  if (2 * kp < nrows) {
    const float* p = src + (long long)(2 * kp) * ld + 8 * cg;
    a0 = *reinterpret_cast<const float4*>(p); a1 = *reinterpret_cast<const float4*>(p + 4);
  }
  if (2 * kp + 1 < nrows) {
    const float* p = src + (long long)(2 * kp + 1) * ld + 8 * cg;
    b0 = *reinterpret_cast<const float4*>(p); b1 = *reinterpret_cast<const float4*>(p + 4);
  }
  float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  float y[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  unsigned* hi32 = reinterpret_cast<unsigned*>(dst);
  unsigned* lo32 = reinterpret_cast<unsigned*>(dst + FA_PLANE);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    unsigned h, l;
    split2(x[c], y[c], h, l);
    const int o = ((8 * cg + c) * FA_LD + 2 * kp) >> 1;
    hi32[o] = h;
    lo32[o] = l;
  }
}

// A operand of a product that sums over the ROWS of an accumulator tile (k-step s: rows 16 s + 8 (j >> 2) + 4 h + (j & 3), the
// order in which the tile's registers 8 s .. 8 s + 7 hold them): lane (r = channel, h) reads two 8-byte runs of Xt[channel][row0 + ...]
__device__ __forceinline__ void load_t_frag(const unsigned short* xt, int row0, int s, int lane, bf16x8& hi, bf16x8& lo) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned short* p = xt + r * FA_LD + row0 + 16 * s + 4 * h;
  const uint2 h0 = *reinterpret_cast<const uint2*>(p), h1 = *reinterpret_cast<const uint2*>(p + 8);
  const uint2 l0 = *reinterpret_cast<const uint2*>(p + FA_PLANE), l1 = *reinterpret_cast<const uint2*>(p + FA_PLANE + 8);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// registers 8 s .. 8 s + 7 of an accumulator tile as the B fragment (hi, lo) of k-step s
__device__ __forceinline__ void acc_frag(const f32x16& x, int s, bf16x8& hi, bf16x8& lo) {
  split8(make_float4(x[8 * s], x[8 * s + 1], x[8 * s + 2], x[8 * s + 3]), make_float4(x[8 * s + 4], x[8 * s + 5], x[8 * s + 6], x[8 * s + 7]), hi, lo);
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// accumulator tile (lane = column c, register = channel) -> X[row0 + c][channel] as four 16-byte stores
__device__ __forceinline__ void store_t(float* base, long long ld, int row0, int nrows, int lane, const f32x16& acc) {
  const int c = lane & 31, hh = lane >> 5;
  if (row0 + c >= nrows) return;
  float* p = base + (long long)(row0 + c) * ld + 4 * hh;
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(p + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

__device__ __forceinline__ bool key_ok(const FlashP& p, const uint8_t* kp, int query, int key) {
  bool ok = key < p.Lk;
  if (p.mask_mode == 1) ok = ok && key <= query + p.causal_offset;
  if (kp) ok = ok && kp[min(key, p.Lk - 1)] == 0;
  return ok;
}

// number of 32-key blocks a 32-query block touches / first query block a key block touches (causal mask)
__device__ __forceinline__ int key_blocks(const FlashP& p, int i) {
  int last = p.Lk - 1;
  if (p.mask_mode == 1) last = min(last, 32 * i + 31 + p.causal_offset);
  return last < 0 ? 0 : last / 32 + 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// forward: O = dropout(softmax(scale Q K^T + mask)) V ; lse = log sum exp of the scaled, masked scores
// ------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(448) flash_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, float* __restrict__ O, float* __restrict__ lse,
                                                         const FlashP p) {
  __shared__ __attribute__((aligned(16))) unsigned short vt[2 * FA_PLANE];
  const int hd = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, i = threadIdx.x >> 6;
  const int c = lane & 31;
  stage_transposed(vt, V + (long long)n * p.bsv + hd * 32, p.ldv, p.Lk);
  __syncthreads();
  if (32 * i >= p.Lq) return;
  const float* Qb = Q + (long long)n * p.bsq + hd * 32;
  const float* Kb = K + (long long)n * p.bsk + hd * 32;
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const Frag2 qf = load_frag(Qb, p.ldq, 32 * i, p.Lq, lane);
  const int nkb = key_blocks(p, i);
  const int query = 32 * i + c;
  f32x16 st[FA_MAXB];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < FA_MAXB; ++j) {
    if (j < nkb) {                                               // wave-uniform
      st[j] = tile_product(load_frag(Kb, p.ldk, 32 * j, p.Lk, lane), qf);      // T = K_j Q_i^T: rows = keys, columns = queries
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = key_ok(p, kp, query, 32 * j + acc_row(r, lane)) ? st[j][r] * p.scale : -INFINITY;
        st[j][r] = s;
        m = fmaxf(m, s);
      }
    }
  }
  m = half_max(m);
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < FA_MAXB; ++j) {
    if (j < nkb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = __expf(st[j][r] - m); st[j][r] = e; l += e; }
    }
  }
  l = half_sum(l);
  const float inv = 1.f / l;                                     // l == 0 (fully masked row): NaN, like torch
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  const uint64_t rbase = (((uint64_t)n * p.H + hd) * p.Lq + (uint64_t)min(query, p.Lq - 1)) * p.Lk;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int j = 0; j < FA_MAXB; ++j) {
    if (j < nkb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = st[j][r] * inv;
        if (p.thresh) pv = cape_keep(seed, step, p.rng_stream, rbase + (uint64_t)(32 * j + acc_row(r, lane)), p.thresh) ? pv * p.inv_keep : 0.f;
        st[j][r] = pv;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {                              // O^T += V_j^T P_j^T
        bf16x8 vhi, vlo, phi, plo;
        load_t_frag(vt, 32 * j, s, lane, vhi, vlo);
        acc_frag(st[j], s, phi, plo);
        acc = mma3(vhi, vlo, phi, plo, acc);
      }
    }
  }
  store_t(O + (long long)n * p.bso + hd * 32, p.ldo, 32 * i, p.Lq, lane, acc);
  if (lane < 32 && query < p.Lq) lse[((long long)n * p.H + hd) * p.Lq + query] = m + __logf(l);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, part 1: dQ and D = rowsum(dO . O) (the softmax-backward constant of each query; read by part 2)
// ------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(448) flash_bwd_dq_kernel(const float* __restrict__ dO, const float* __restrict__ Q,
                                                            const float* __restrict__ K, const float* __restrict__ V,
                                                            const float* __restrict__ O, const float* __restrict__ lse,
                                                            float* __restrict__ dQ, float* __restrict__ Dws, const FlashP p) {
  __shared__ __attribute__((aligned(16))) unsigned short kt[2 * FA_PLANE];
  const int hd = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, i = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const float* Kb = K + (long long)n * p.bsk + hd * 32;
  stage_transposed(kt, Kb, p.ldk, p.Lk);
  __syncthreads();
  if (32 * i >= p.Lq) return;
  const float* Vb = V + (long long)n * p.bsv + hd * 32;
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const int query = 32 * i + c, qc = min(query, p.Lq - 1);
  const Frag2 qf = load_frag(Q + (long long)n * p.bsq + hd * 32, p.ldq, 32 * i, p.Lq, lane);
  // dO fragment (B operand of dP^T = V_j dO_i^T) and, from the same 16 channels of this lane, its share of D
  Frag2 gf;
  float D;
  {
    const float* gp = dO + (long long)n * p.bso + hd * 32 + (long long)qc * p.ldo + 8 * h;
    const float* op = O + (long long)n * p.bso + hd * 32 + (long long)qc * p.ldo + 8 * h;
    float d = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float4 g0 = *reinterpret_cast<const float4*>(gp + 16 * s), g1 = *reinterpret_cast<const float4*>(gp + 16 * s + 4);
      const float4 o0 = *reinterpret_cast<const float4*>(op + 16 * s), o1 = *reinterpret_cast<const float4*>(op + 16 * s + 4);
      d += g0.x * o0.x + g0.y * o0.y + g0.z * o0.z + g0.w * o0.w + g1.x * o1.x + g1.y * o1.y + g1.z * o1.z + g1.w * o1.w;
      split8(g0, g1, gf.hi[s], gf.lo[s]);
    }
    D = half_sum(d);
  }
  const long long stat = ((long long)n * p.H + hd) * p.Lq + qc;
  if (lane < 32 && query < p.Lq) Dws[stat] = D;
  const float lse_c = lse[stat];
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  const uint64_t rbase = (uint64_t)stat * p.Lk;
  const int nkb = key_blocks(p, i);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j = 0; j < nkb; ++j) {
    f32x16 t = tile_product(load_frag(Kb, p.ldk, 32 * j, p.Lk, lane), qf);          // S^T
    const f32x16 dp = tile_product(load_frag(Vb, p.ldv, 32 * j, p.Lk, lane), gf);   // dP^T = V_j dO_i^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * j + acc_row(r, lane);
      const float pr = key_ok(p, kp, query, key) ? __expf(t[r] * p.scale - lse_c) : 0.f;
      float g = dp[r];
      if (p.thresh) g = cape_keep(seed, step, p.rng_stream, rbase + (uint64_t)key, p.thresh) ? g * p.inv_keep : 0.f;
      t[r] = pr * (g - D) * p.scale;                             // dS^T
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                // dQ^T += K_j^T dS^T
      bf16x8 khi, klo, shi, slo;
      load_t_frag(kt, 32 * j, s, lane, khi, klo);
      acc_frag(t, s, shi, slo);
      acc = mma3(khi, klo, shi, slo, acc);
    }
  }
  store_t(dQ + (long long)n * p.bsq + hd * 32, p.ldq, 32 * i, p.Lq, lane, acc);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, part 2: dK and dV of one 32-key block per wave
// ------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(448) flash_bwd_dkv_kernel(const float* __restrict__ dO, const float* __restrict__ Q,
                                                             const float* __restrict__ K, const float* __restrict__ V,
                                                             const float* __restrict__ lse, const float* __restrict__ Dws,
                                                             float* __restrict__ dK, float* __restrict__ dV, const FlashP p) {
  __shared__ __attribute__((aligned(16))) unsigned short qt[2 * FA_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned short gt[2 * FA_PLANE];
  __shared__ float s_lse[32 * FA_MAXB], s_D[32 * FA_MAXB];
  const int hd = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int c = lane & 31;
  const float* Qb = Q + (long long)n * p.bsq + hd * 32;
  const float* Gb = dO + (long long)n * p.bso + hd * 32;
  stage_transposed(qt, Qb, p.ldq, p.Lq);
  stage_transposed(gt, Gb, p.ldo, p.Lq);
  if (threadIdx.x < 32 * FA_MAXB) {
    const long long stat = ((long long)n * p.H + hd) * p.Lq + min((int)threadIdx.x, p.Lq - 1);
    s_lse[threadIdx.x] = lse[stat];
    s_D[threadIdx.x] = Dws[stat];
  }
  __syncthreads();
  if (32 * j >= p.Lk) return;
  const uint8_t* kp = p.mask_mode == 2 ? p.kpm + (long long)n * p.Lk : nullptr;
  const Frag2 kf = load_frag(K + (long long)n * p.bsk + hd * 32, p.ldk, 32 * j, p.Lk, lane);
  const Frag2 vf = load_frag(V + (long long)n * p.bsv + hd * 32, p.ldv, 32 * j, p.Lk, lane);
  const int key = 32 * j + c;
  uint64_t seed = 0, step = 0;
  if (p.thresh) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  const uint64_t hbase = ((uint64_t)n * p.H + hd) * p.Lq;
  const int nrb = (p.Lq + 31) / 32;
  int i0 = 0;                                                    // first query block that sees a key of this block
  if (p.mask_mode == 1) i0 = max(0, 32 * j - p.causal_offset) / 32;
  f32x16 acc_k, acc_v;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_k[r] = 0.f; acc_v[r] = 0.f; }
  for (int i = i0; i < nrb; ++i) {
    f32x16 s = tile_product(load_frag(Qb, p.ldq, 32 * i, p.Lq, lane), kf);          // S = Q_i K_j^T: rows = queries, columns = keys
    f32x16 dp = tile_product(load_frag(Gb, p.ldo, 32 * i, p.Lq, lane), vf);         // dP = dO_i V_j^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = 32 * i + acc_row(r, lane);
      const bool ok = qrow < p.Lq && key_ok(p, kp, qrow, key);
      const float pr = ok ? __expf(s[r] * p.scale - s_lse[qrow]) : 0.f;
      float keep = 1.f;
      if (p.thresh) keep = cape_keep(seed, step, p.rng_stream, (hbase + (uint64_t)min(qrow, p.Lq - 1)) * p.Lk + (uint64_t)min(key, p.Lk - 1), p.thresh) ? p.inv_keep : 0.f;
      s[r] = pr * (dp[r] * keep - s_D[qrow]) * p.scale;          // dS
      dp[r] = pr * keep;                                         // dropped P
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ahi, alo, bhi, blo;
      load_t_frag(gt, 32 * i, ks, lane, ahi, alo);               // dV^T += dO_i^T Pd
      acc_frag(dp, ks, bhi, blo);
      acc_v = mma3(ahi, alo, bhi, blo, acc_v);
      load_t_frag(qt, 32 * i, ks, lane, ahi, alo);               // dK^T += Q_i^T dS
      acc_frag(s, ks, bhi, blo);
      acc_k = mma3(ahi, alo, bhi, blo, acc_k);
    }
  }
  store_t(dK + (long long)n * p.bsk + hd * 32, p.ldk, 32 * j, p.Lk, lane, acc_k);
  store_t(dV + (long long)n * p.bsv + hd * 32, p.ldv, 32 * j, p.Lk, lane, acc_v);
}

int fill(FlashP& p, long long ldq, long long ldk, long long ldv, long long ldo, long long bsq, long long bsk, long long bsv, long long bso,
         int N, int H, int Lq, int Lk, float scale, int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p,
         const uint64_t* rng_state, uint32_t rng_stream, const char* who) {
  CAPE_REQUIRE(N >= 1 && H >= 1 && N <= 65535 && H <= 65535, "%s: bad batch / head count", who);
  CAPE_REQUIRE(Lq >= 1 && Lk >= 1 && Lq <= 32 * FA_MAXB && Lk <= 32 * FA_MAXB, "%s: sequence lengths must be in 1..%d (Lq=%d, Lk=%d)", who, 32 * FA_MAXB, Lq, Lk);
  CAPE_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && bsq % 4 == 0 && bsk % 4 == 0 && bsv % 4 == 0 && bso % 4 == 0,
               "%s: strides must be multiples of 4 floats", who);
  CAPE_REQUIRE(mask_mode >= 0 && mask_mode <= 2 && (mask_mode != 2 || kpm), "%s: bad mask mode", who);
  CAPE_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || rng_state), "%s: dropout needs rng_state and p < 1", who);
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
  p.N = N; p.H = H; p.Lq = Lq; p.Lk = Lk; p.scale = scale; p.mask_mode = mask_mode; p.causal_offset = causal_offset; p.kpm = kpm;
  p.thresh = dropout_p > 0.f ? cape_drop_threshold(dropout_p) : 0u;
  p.inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  p.rng_state = rng_state; p.rng_stream = rng_stream;
  return 0;
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int cape_flash_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, long long ldq, long long ldk,
                                   long long ldv, long long ldo, long long bsq, long long bsk, long long bsv, long long bso, int N, int H,
                                   int Lq, int Lk, float scale, int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p,
                                   const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(Q && K && V && O && lse, "cape_flash_attn_fwd: null pointer");
  CAPE_REQUIRE(al16(Q) && al16(K) && al16(V) && al16(O), "cape_flash_attn_fwd: operands must be 16-byte aligned");
  FlashP p;
  if (fill(p, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, N, H, Lq, Lk, scale, mask_mode, causal_offset, kpm, dropout_p, rng_state, rng_stream,
           "cape_flash_attn_fwd")) return 1;
  hipLaunchKernelGGL(flash_fwd_kernel, dim3((unsigned)H, (unsigned)N), dim3(448), 0, as_stream(stream), Q, K, V, O, lse, p);
  CAPE_LAUNCH_CHECK("cape_flash_attn_fwd");
  return 0;
}

extern "C" int cape_flash_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* O, const float* lse,
                                   float* dQ, float* dK, float* dV, float* d_ws, long long ldq, long long ldk, long long ldv, long long ldo,
                                   long long bsq, long long bsk, long long bsv, long long bso, int N, int H, int Lq, int Lk, float scale,
                                   int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
                                   uint32_t rng_stream, cape_stream_t stream) {
  CAPE_REQUIRE(dO && Q && K && V && O && lse && dQ && dK && dV && d_ws, "cape_flash_attn_bwd: null pointer");
  CAPE_REQUIRE(al16(dO) && al16(Q) && al16(K) && al16(V) && al16(O) && al16(dQ) && al16(dK) && al16(dV),
               "cape_flash_attn_bwd: operands must be 16-byte aligned");
  FlashP p;
  if (fill(p, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, N, H, Lq, Lk, scale, mask_mode, causal_offset, kpm, dropout_p, rng_state, rng_stream,
           "cape_flash_attn_bwd")) return 1;
  const dim3 grid((unsigned)H, (unsigned)N), block(448);
  hipLaunchKernelGGL(flash_bwd_dq_kernel, grid, block, 0, as_stream(stream), dO, Q, K, V, O, lse, dQ, d_ws, p);
  hipLaunchKernelGGL(flash_bwd_dkv_kernel, grid, block, 0, as_stream(stream), dO, Q, K, V, lse, d_ws, dK, dV, p);
  CAPE_LAUNCH_CHECK("cape_flash_attn_bwd");
  return 0;
}
