// msda.hip -- multi-scale deformable attention core for gfx950 (wave64).
//
// Forward: one wave per query token.  lane = head*8 + c4: the 8 lanes of a head own 4 channels each
// (16-byte loads; a tap of one head is one 128-byte line), each lane also owns two of the head's
// L*P (<= 16) samples for the softmax, which is finished with three xor-shuffles inside the
// 8-lane group.  The value tensor of one image (S x 256 fp32 = 1.4 MB at 256x256) stays in the XCD's
// L2; blocks of one image are steered to one XCD (blockIdx % 8 == image % 8 ordering).
//
// Backward: one wave per (query, head pair): lane = half*32 + channel so that every atomic
// wave-instruction on d_value is two full 128-byte segments (the shape the memory-side float atomics
// run at full rate for, MI355X_MICROARCH.md "Global float atomics").
#include <stdlib.h>
#include "common.h"

namespace {

struct Levels {
  int H[4], W[4], start[4];
};

constexpr int HEADS = 8, HD = 32, CH = HEADS * HD;  // 256

// per-lane level lookups as select chains (dynamic indexing of a kernel-argument array would go to scratch)
__device__ __forceinline__ int sel4(const int (&a)[4], int l) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); }

__global__ void __launch_bounds__(256) msda_fwd_kernel(const float* __restrict__ value, const float* __restrict__ offw,
                                                        const float* __restrict__ ref, float* __restrict__ out,
                                                        Levels lv, int N, int S, int Lq, int L, int P, int blocks_per_image) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // image-major block order with XCD steering: block b -> (image, chunk)
  int n, chunk;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, loc = b >> 3;          // blocks sharing b&7 share an XCD
    const int imgs_per_x = (N + 7) >> 3;
    const int ii = loc / blocks_per_image;
    chunk = loc - ii * blocks_per_image;
    n = ii * 8 + xcd;
    if (ii >= imgs_per_x || n >= N) return;
  }
  const int q = chunk * 4 + wv;
  if (q >= Lq) return;
  const int LP = L * P;
  const int h = lane >> 3, c4 = lane & 7;
  const long long qrow = (long long)n * Lq + q;
  const float* ow = offw + qrow * (HEADS * LP * 3);
  // the row holds [HEADS*LP*2 offsets | HEADS*LP logits]
  const float* offs = ow + h * LP * 2;
  const float* logit = ow + HEADS * LP * 2 + h * LP;
  float px[2], py[2], lg[2];
  int lev[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = c4 + 8 * s;
    if (j < LP) {
      const int l = j / P;
      lev[s] = l;
      const float rx = ref[(qrow * L + l) * 2 + 0], ry = ref[(qrow * L + l) * 2 + 1];
      const float ox = offs[j * 2 + 0], oy = offs[j * 2 + 1];
      const float Wf = (float)sel4(lv.W, l), Hf = (float)sel4(lv.H, l);
      px[s] = (rx + ox / Wf) * Wf - 0.5f;
      py[s] = (ry + oy / Hf) * Hf - 0.5f;
      lg[s] = logit[j];
    } else {
      lev[s] = 0; px[s] = 0.f; py[s] = 0.f; lg[s] = -INFINITY;
    }
  }
  float mx = fmaxf(lg[0], lg[1]);
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
  float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx);
  float sm = e0 + e1;
  sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
  const float inv = 1.f / sm;
  const float aw[2] = {e0 * inv, e1 * inv};

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* vbase = value + (long long)n * S * CH + h * HD + c4 * 4;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = jj + 8 * s;
      if (j >= LP) break;
      const int src = (lane & ~7) | jj;
      const float x = __shfl(px[s], src, 64), y = __shfl(py[s], src, 64), a = __shfl(aw[s], src, 64);
      const int l = __shfl(lev[s], src, 64);
      const int W = sel4(lv.W, l), H = sel4(lv.H, l);
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      const float* vl = vbase + (long long)sel4(lv.start, l) * CH;
      const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
      const bool ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
      float4 v;
      if (ya && xa) { v = *reinterpret_cast<const float4*>(vl + (long long)(y0 * W + x0) * CH); const float w = a * (1.f - fx) * (1.f - fy);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w; }
      if (ya && xb) { v = *reinterpret_cast<const float4*>(vl + (long long)(y0 * W + x0 + 1) * CH); const float w = a * fx * (1.f - fy);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w; }
      if (yb && xa) { v = *reinterpret_cast<const float4*>(vl + (long long)((y0 + 1) * W + x0) * CH); const float w = a * (1.f - fx) * fy;
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w; }
      if (yb && xb) { v = *reinterpret_cast<const float4*>(vl + (long long)((y0 + 1) * W + x0 + 1) * CH); const float w = a * fx * fy;
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w; }
    }
  }
  *reinterpret_cast<float4*>(out + qrow * CH + h * HD + c4 * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes of a half wave
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
  return v;
}

__global__ void __launch_bounds__(256) msda_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ value,
                                                        const float* __restrict__ offw, const float* __restrict__ ref,
                                                        float* __restrict__ d_value, float* __restrict__ d_offw,
                                                        float* __restrict__ d_ref, Levels lv, int N, int S, int Lq, int L,
                                                        int P, long long total_waves) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * 4 + wv;     // = (n*Lq + q)*4 + headpair
  if (wid >= total_waves) return;
  const int pair = (int)(wid & 3);
  const long long qrow = wid >> 2;
  const int n = (int)(qrow / Lq);
  const int hh = lane >> 5, c = lane & 31;
  const int h = pair * 2 + hh;
  const int LP = L * P;
  const int j = c & 15;                                      // the sample this lane parameterises
  const int rowlen = HEADS * LP * 3;
  const float* ow = offw + qrow * rowlen;
  float px = 0.f, py = 0.f, lg = -INFINITY;
  int l_own = 0;
  if (j < LP) {
    l_own = j / P;
    const float rx = ref[(qrow * L + l_own) * 2 + 0], ry = ref[(qrow * L + l_own) * 2 + 1];
    const float ox = ow[(h * LP + j) * 2 + 0], oy = ow[(h * LP + j) * 2 + 1];
    const float Wf = (float)sel4(lv.W, l_own), Hf = (float)sel4(lv.H, l_own);
    px = (rx + ox / Wf) * Wf - 0.5f;
    py = (ry + oy / Hf) * Hf - 0.5f;
    lg = ow[HEADS * LP * 2 + h * LP + j];
  }
  float mx = lg;
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
  const float e = __expf(lg - mx);
  float sm = e;
  sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64); sm += __shfl_xor(sm, 8, 64);
  const float aw = e / sm;                                  // lanes c and c+16 hold the same sample

  const float go = d_out[qrow * CH + h * HD + c];
  const float* vbase = value + (long long)n * S * CH + h * HD + c;
  float* dvbase = d_value + (long long)n * S * CH + h * HD + c;
  float my_daw = 0.f, my_dpx = 0.f, my_dpy = 0.f;
  for (int jj = 0; jj < LP; ++jj) {
    const int src = (lane & 32) | jj;
    const float x = __shfl(px, src, 64), y = __shfl(py, src, 64), a = __shfl(aw, src, 64);
    const int l = __shfl(l_own, src, 64);
    const int W = sel4(lv.W, l), H = sel4(lv.H, l);
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const int x0 = (int)xf, y0 = (int)yf;
    const long long lofs = (long long)sel4(lv.start, l) * CH;
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
    const bool ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    const float ga = go * a;
    if (ya && xa) { const long long o = lofs + (long long)(y0 * W + x0) * CH; v00 = vbase[o]; atomicAdd(dvbase + o, ga * (1.f - fx) * (1.f - fy)); }
    if (ya && xb) { const long long o = lofs + (long long)(y0 * W + x0 + 1) * CH; v01 = vbase[o]; atomicAdd(dvbase + o, ga * fx * (1.f - fy)); }
    if (yb && xa) { const long long o = lofs + (long long)((y0 + 1) * W + x0) * CH; v10 = vbase[o]; atomicAdd(dvbase + o, ga * (1.f - fx) * fy); }
    if (yb && xb) { const long long o = lofs + (long long)((y0 + 1) * W + x0 + 1) * CH; v11 = vbase[o]; atomicAdd(dvbase + o, ga * fx * fy); }
    const float samp = (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
    const float dsx = (1.f - fy) * (v01 - v00) + fy * (v11 - v10);
    const float dsy = (1.f - fx) * (v10 - v00) + fx * (v11 - v01);
    const float r_aw = half_sum32(go * samp);
    const float r_px = half_sum32(ga * dsx);
    const float r_py = half_sum32(ga * dsy);
    if (j == jj) { my_daw = r_aw; my_dpx = r_px; my_dpy = r_py; }
  }
  // softmax backward over the head's LP samples (lanes 0..15 of each half; 16..31 are duplicates)
  float dot = aw * my_daw;
  dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);
  const float dlogit = aw * (my_daw - dot);
  float* dow = d_offw + qrow * rowlen;
  if (c < 16 && j < LP) {
    dow[(h * LP + j) * 2 + 0] = my_dpx;      // d px / d offset_x = 1
    dow[(h * LP + j) * 2 + 1] = my_dpy;
    dow[HEADS * LP * 2 + h * LP + j] = dlogit;
  }
  if (d_ref) {
    // d px / d ref_x = W_l : sum over the P points of a level (and over heads via atomics)
    float rx = (c < 16 && j < LP) ? my_dpx * (float)sel4(lv.W, l_own) : 0.f;
    float ry = (c < 16 && j < LP) ? my_dpy * (float)sel4(lv.H, l_own) : 0.f;
    // segmented sum over the P consecutive lanes of a level (P is 4 here; general P handled serially)
    if (P == 4) {
      rx += __shfl_xor(rx, 1, 64); rx += __shfl_xor(rx, 2, 64);
      ry += __shfl_xor(ry, 1, 64); ry += __shfl_xor(ry, 2, 64);
      if (c < 16 && j < LP && (j & 3) == 0) {
        atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 0], rx);
        atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 1], ry);
      }
    } else if (c < 16 && j < LP) {
      atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 0], rx);
      atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 1], ry);
    }
  }
}



// ---------------------------------------------------------------------------------------------
// backward, split form (the training path).  The two halves of the backward share nothing but the sampling geometry,
// so they are separate kernels and neither touches a memory-side atomic:
//
//   msda_bwd_offw_kernel   d_offw / d_ref.  Gather-only, one wave per query, the forward's lane layout
//                          (lane = head*8 + c4, 4 channels per lane, a tap of one head = one 128-byte line).
//                          Each lane owns two of its head's samples: it computes their geometry once and parks it in
//                          a wave-private LDS record; the 8 lanes of the head then walk the 16 records, gather the
//                          corners, and reduce go.samp / go.dsx / go.dsy over the head with three DPP steps.
//
//   msda_bwd_value_kernel  d_value.  No gathers at all: d_value[n, pix, h, c] = sum over taps of aw * w_corner * go.
//                          One block per (image, head, 8-channel group): the slab (S+1) x 8 accumulators lives in LDS
//                          and is written back with plain stores (no zero fill of d_value needed).  The accumulators
//                          are fp64 on purpose: measured on MI355X (tools/lab/lds_atomic_bench.hip), a conflict-free
//                          ds_add_f32 wave-instruction costs ~190 clocks (the lanes are serialised), ds_add_f64 8,
//                          ds_add_u32 4 -- fp32 LDS atomics made the slab form 2x SLOWER than memory-side atomics,
//                          fp64 ones make it several times faster (and the sum is rounded to fp32 once, at the end).
//                          Memory-side float atomics run at ~1.3 TB/s chip-wide, which bounded msda_bwd_kernel at
//                          1.8 ms per encoder layer (N = 32, S = Lq = 1360).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// butterflies inside a DPP row: quad xor 1 (0xB1), quad xor 2 (0x4E), half-row mirror (0x141), row mirror (0x140)
__device__ __forceinline__ float sum8(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); return v; }
__device__ __forceinline__ float max8(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}
__device__ __forceinline__ float sum16(float v) { v = sum8(v); v += dpp_f<0x140>(v); return v; }
__device__ __forceinline__ float max16(float v) { v = max8(v); v = fmaxf(v, dpp_f<0x140>(v)); return v; }

// sampling geometry of one (query, head, sample): corner pixel ids (S = tap outside the level) and fractions
struct Tap { unsigned i00, i01, i10, i11; float fx, fy; };
__device__ __forceinline__ Tap make_tap(float px, float py, int W, int H, int start, bool live, int S) {
  const float xf = floorf(px), yf = floorf(py);
  Tap t;
  t.fx = px - xf; t.fy = py - yf;
  const int x0 = (int)xf, y0 = (int)yf;
  const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
  const bool ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
  const int base = start + y0 * W + x0;
  t.i00 = (live && ya && xa) ? (unsigned)base : (unsigned)S;
  t.i01 = (live && ya && xb) ? (unsigned)(base + 1) : (unsigned)S;
  t.i10 = (live && yb && xa) ? (unsigned)(base + W) : (unsigned)S;
  t.i11 = (live && yb && xb) ? (unsigned)(base + W + 1) : (unsigned)S;
  return t;
}

// Forward, record form (the training and decode path): same lane layout as msda_fwd_kernel (lane = head*8 + c4), but each
// lane computes the geometry of its own two samples once and parks {corner ids, aw * bilinear weights} in a wave-private
// LDS record (weights of taps outside a level are 0, their ids clamped: the walk below is branch-free), and the 8 lanes of
// a head then walk the 16 records with the gathers of sample j+PF issued before sample j is consumed.  The shuffle form
// re-derived the geometry in every lane and waited for each sample's four dependent L2 reads in turn (27 us for the 32
// single-query rows of a decode step, 88 us per encoder layer).
__global__ void __launch_bounds__(256) msda_fwd_rec_kernel(const float* __restrict__ value, const float* __restrict__ offw,
                                                            const float* __restrict__ ref, float* __restrict__ out, Levels lv,
                                                            int N, int S, int Lq, int L, int P, int blocks_per_image) {
  __shared__ uint2 rec_ids[4][HEADS * 16];
  __shared__ float4 rec_w[4][HEADS * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int n, chunk;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, loc = b >> 3;
    const int ii = loc / blocks_per_image;
    chunk = loc - ii * blocks_per_image;
    n = ii * 8 + xcd;
    if (n >= N) return;
  }
  const int q = chunk * 4 + wv;
  if (q >= Lq) return;                                           // wave-uniform
  const int LP = L * P;
  const int h = lane >> 3, i = lane & 7;
  const long long qrow = (long long)n * Lq + q;
  const float* ow = offw + qrow * (HEADS * LP * 3);
  float px[2], py[2], lg[2];
  int W[2], H[2], st[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = i + 8 * s;
    px[s] = 0.f; py[s] = 0.f; lg[s] = -INFINITY; W[s] = 1; H[s] = 1; st[s] = 0;
    if (j < LP) {
      const int l = j / P;
      W[s] = sel4(lv.W, l); H[s] = sel4(lv.H, l); st[s] = sel4(lv.start, l);
      const float2 rr = *reinterpret_cast<const float2*>(ref + (qrow * L + l) * 2);
      const float2 oo = *reinterpret_cast<const float2*>(ow + (h * LP + j) * 2);
      const float Wf = (float)W[s], Hf = (float)H[s];
      px[s] = (rr.x + oo.x / Wf) * Wf - 0.5f;
      py[s] = (rr.y + oo.y / Hf) * Hf - 0.5f;
      lg[s] = ow[HEADS * LP * 2 + h * LP + j];
    }
  }
  const float mx = max8(fmaxf(lg[0], lg[1]));
  const float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx);
  const float inv = 1.f / sum8(e0 + e1);
  const float aw[2] = {e0 * inv, e1 * inv};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = i + 8 * s;
    const Tap t = make_tap(px[s], py[s], W[s], H[s], st[s], j < LP, S);
    const unsigned cl = (unsigned)(S - 1);
    float4 w = make_float4(aw[s] * (1.f - t.fx) * (1.f - t.fy), aw[s] * t.fx * (1.f - t.fy), aw[s] * (1.f - t.fx) * t.fy,
                           aw[s] * t.fx * t.fy);
    if (t.i00 >= (unsigned)S) w.x = 0.f;
    if (t.i01 >= (unsigned)S) w.y = 0.f;
    if (t.i10 >= (unsigned)S) w.z = 0.f;
    if (t.i11 >= (unsigned)S) w.w = 0.f;
    rec_ids[wv][h * 16 + j] = make_uint2(min(t.i00, cl) | (min(t.i01, cl) << 16), min(t.i10, cl) | (min(t.i11, cl) << 16));
    rec_w[wv][h * 16 + j] = w;
  }
  const float* vb = value + (long long)n * S * CH + h * HD + i * 4;
  const uint2* rid = &rec_ids[wv][h * 16];
  const float4* rw = &rec_w[wv][h * 16];
  constexpr int PF = 4;
  float4 v[PF + 1][4];
  auto fetch = [&](int j, int slot) {
    const uint2 ids = rid[j];                                    // same wave wrote it: ordered by lgkmcnt, no barrier
    v[slot][0] = *reinterpret_cast<const float4*>(vb + (ids.x & 0xFFFFu) * (unsigned)CH);
    v[slot][1] = *reinterpret_cast<const float4*>(vb + (ids.x >> 16) * (unsigned)CH);
    v[slot][2] = *reinterpret_cast<const float4*>(vb + (ids.y & 0xFFFFu) * (unsigned)CH);
    v[slot][3] = *reinterpret_cast<const float4*>(vb + (ids.y >> 16) * (unsigned)CH);
  };
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < LP) fetch(j, j);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j >= LP) continue;                                       // uniform; the loop stays fully unrolled
    if (j + PF < LP) fetch(j + PF, (j + PF) % (PF + 1));
    const int slot = j % (PF + 1);
    const float4 w = rw[j];
    acc.x += w.x * v[slot][0].x + w.y * v[slot][1].x + w.z * v[slot][2].x + w.w * v[slot][3].x;
    acc.y += w.x * v[slot][0].y + w.y * v[slot][1].y + w.z * v[slot][2].y + w.w * v[slot][3].y;
    acc.z += w.x * v[slot][0].z + w.y * v[slot][1].z + w.z * v[slot][2].z + w.w * v[slot][3].z;
    acc.w += w.x * v[slot][0].w + w.y * v[slot][1].w + w.z * v[slot][2].w + w.w * v[slot][3].w;
  }
  *reinterpret_cast<float4*>(out + qrow * CH + h * HD + i * 4) = acc;
}

__global__ void __launch_bounds__(256) msda_bwd_offw_kernel(const float* __restrict__ d_out, const float* __restrict__ value,
                                                             const float* __restrict__ offw, const float* __restrict__ ref,
                                                             float* __restrict__ d_offw, float* __restrict__ d_ref, Levels lv,
                                                             int N, int S, int Lq, int L, int P, int blocks_per_image) {
  __shared__ uint4 recs[4][HEADS * 16];                          // per wave: [head][sample] {i00|i01<<16, i10|i11<<16, fx, fy}
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int n, chunk;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, loc = b >> 3;                         // blocks sharing b&7 share an XCD: one image, one L2
    const int ii = loc / blocks_per_image;
    chunk = loc - ii * blocks_per_image;
    n = ii * 8 + xcd;
    if (n >= N) return;
  }
  const int q = chunk * 4 + wv;
  if (q >= Lq) return;                                           // wave-uniform: EXEC stays full for the DPP butterflies
  const int LP = L * P;
  const int h = lane >> 3, i = lane & 7;
  const long long qrow = (long long)n * Lq + q;
  const int rowlen = HEADS * LP * 3;
  const float* ow = offw + qrow * rowlen;
  float lg[2], aw[2];
  int lev[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = i + 8 * s;
    const bool live = j < LP;
    float px = 0.f, py = 0.f;
    int W = 1, H = 1, st = 0;
    lg[s] = -INFINITY; lev[s] = 0;
    if (live) {
      const int l = j / P;
      lev[s] = l;
      W = sel4(lv.W, l); H = sel4(lv.H, l); st = sel4(lv.start, l);
      const float2 rr = *reinterpret_cast<const float2*>(ref + (qrow * L + l) * 2);
      const float2 oo = *reinterpret_cast<const float2*>(ow + (h * LP + j) * 2);
      const float Wf = (float)W, Hf = (float)H;
      px = (rr.x + oo.x / Wf) * Wf - 0.5f;
      py = (rr.y + oo.y / Hf) * Hf - 0.5f;
      lg[s] = ow[HEADS * LP * 2 + h * LP + j];
    }
    const Tap t = make_tap(px, py, W, H, st, live, S);
    recs[wv][h * 16 + j] = make_uint4(t.i00 | (t.i01 << 16), t.i10 | (t.i11 << 16), __float_as_uint(t.fx), __float_as_uint(t.fy));
  }
  {
    const float mx = max8(fmaxf(lg[0], lg[1]));
    const float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx);
    const float inv = 1.f / sum8(e0 + e1);
    aw[0] = e0 * inv; aw[1] = e1 * inv;
  }
  const float4 go = *reinterpret_cast<const float4*>(d_out + qrow * CH + h * HD + i * 4);
  const float* vb = value + (long long)n * S * CH + h * HD + i * 4;
  float daw[2] = {0.f, 0.f}, dpx[2] = {0.f, 0.f}, dpy[2] = {0.f, 0.f};
  const uint4* myrec = &recs[wv][h * 16];
  constexpr int PF = 2;                                          // records / gathers in flight ahead of the consumer
  uint4 ra[PF + 1];
  float4 v[PF + 1][4];
  auto fetch = [&](int j, int slot) {
    ra[slot] = myrec[j];                                         // same wave wrote it: ordered by lgkmcnt, no barrier
    const unsigned ids[4] = {ra[slot].x & 0xFFFFu, ra[slot].x >> 16, ra[slot].y & 0xFFFFu, ra[slot].y >> 16};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned idc = min(ids[k], (unsigned)(S - 1));
      const float4 x = *reinterpret_cast<const float4*>(vb + idc * (unsigned)CH);
      v[slot][k] = ids[k] < (unsigned)S ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < LP) fetch(j, j);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j >= LP) continue;                                       // uniform; the loop stays fully unrolled
    if (j + PF < LP) fetch(j + PF, (j + PF) % (PF + 1));
    const int slot = j % (PF + 1);
    const float fx = __uint_as_float(ra[slot].z), fy = __uint_as_float(ra[slot].w);
    const float4 v00 = v[slot][0], v01 = v[slot][1], v10 = v[slot][2], v11 = v[slot][3];
    float p_aw = 0.f, p_x = 0.f, p_y = 0.f;
#define CAPE_TAP_CH(C)                                                             \
    {                                                                              \
      const float dt = v01.C - v00.C, db = v11.C - v10.C;                          \
      const float top = v00.C + fx * dt, bot = v10.C + fx * db;                    \
      p_aw += go.C * (top + fy * (bot - top));                                     \
      p_x += go.C * (dt + fy * (db - dt));                                         \
      p_y += go.C * (bot - top);                                                   \
    }
    CAPE_TAP_CH(x) CAPE_TAP_CH(y) CAPE_TAP_CH(z) CAPE_TAP_CH(w)
#undef CAPE_TAP_CH
    const float r_aw = sum8(p_aw), r_x = sum8(p_x), r_y = sum8(p_y);
    if (i == (j & 7)) { daw[j >> 3] = r_aw; dpx[j >> 3] = r_x; dpy[j >> 3] = r_y; }
  }
  // softmax backward over the head's samples, then the outputs of this lane's two samples
  const float dot = sum8(aw[0] * daw[0] + aw[1] * daw[1]);
  float* dow = d_offw + qrow * rowlen;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = i + 8 * s;
    const bool live = j < LP;
    const float gx = aw[s] * dpx[s], gy = aw[s] * dpy[s];        // d px / d offset_x = 1 (the 1/W of the offset and the W of px cancel)
    if (live) {
      *reinterpret_cast<float2*>(dow + (h * LP + j) * 2) = make_float2(gx, gy);
      dow[HEADS * LP * 2 + h * LP + j] = aw[s] * (daw[s] - dot);
    }
    if (d_ref) {
      // d px / d ref_x = W_l: sum over the P points of a level here, over heads through the atomics
      float rx = live ? gx * (float)sel4(lv.W, lev[s]) : 0.f;
      float ry = live ? gy * (float)sel4(lv.H, lev[s]) : 0.f;
      if (P == 4) {
        // the wave owns the query: sum over the 4 points of a level (quad DPP), then over the 8 heads (lanes 8 apart) -- one plain
        // store per (level, axis), no atomics and no zero fill (round 2: 64 atomics per query onto 8 addresses: the decoder's
        // backward, where the reference points are learned, ran 54 us against 24 us for the forward)
        rx += dpp_f<0xB1>(rx); rx += dpp_f<0x4E>(rx);
        ry += dpp_f<0xB1>(ry); ry += dpp_f<0x4E>(ry);
        rx += __shfl_xor(rx, 8); rx += __shfl_xor(rx, 16); rx += __shfl_xor(rx, 32);
        ry += __shfl_xor(ry, 8); ry += __shfl_xor(ry, 16); ry += __shfl_xor(ry, 32);
        if (live && h == 0 && (i & 3) == 0) *reinterpret_cast<float2*>(&d_ref[(qrow * L + lev[s]) * 2]) = make_float2(rx, ry);
      } else if (live) {
        atomicAdd(&d_ref[(qrow * L + lev[s]) * 2 + 0], rx);
        atomicAdd(&d_ref[(qrow * L + lev[s]) * 2 + 1], ry);
      }
    }
  }
}

// channels per block of the value kernel: 8 while the slab (S+1) x 8 fp64 fits in LDS (S <= ~2100: up to 320x320 images),
// then 4 (384x384: S = 3060) and 2 (512x512: S = 5440); the slab row stride equals the group width (a padded stride of
// 9 or 10 doubles measured 10-13 % slower)
struct ValRec { uint2 ids; float4 w; };                          // 24 bytes: corner ids + aw * bilinear weights

template <int VC>
__global__ void __launch_bounds__(1024) msda_bwd_value_kernel(const float* __restrict__ d_out, const float* __restrict__ offw,
                                                              const float* __restrict__ ref, float* __restrict__ d_value,
                                                              Levels lv, int N, int S, int Lq, int L, int P) {
  constexpr int VS = VC, GROUPS = HD / VC, BPI = HEADS * GROUPS;   // slab row stride, channel groups per head, blocks per image
  constexpr int SLOTS = 64 / VC;                                  // samples a wave advances per step; VC steps cover its 64 records
  extern __shared__ __attribute__((aligned(16))) double slab[];  // [(S + 1)][VS]; row S swallows taps outside a level
  __shared__ uint2 rec_ids[16][64];                              // per wave: [query g (4)][sample j (16)]
  __shared__ float4 rec_w[16][64];
  int n, sub;
  {
    const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;         // the BPI blocks of an image share an XCD
    n = (loc / BPI) * 8 + xcd;
    sub = loc % BPI;
    if (n >= N) return;
  }
  const int h = sub / GROUPS, grp = sub % GROUPS;
  for (int k = threadIdx.x; k < (S + 1) * VS; k += 1024) slab[k] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;                        // phase 1: (query, sample)
  const int c = lane & (VC - 1), slot = lane / VC;               // phase 2: channel c; records slot*VC .. slot*VC+VC-1
  static_assert(SLOTS * VC == 64 && 16 % VC == 0, "a slot's records must stay inside one query");
  const int LP = L * P;
  const int rowlen = HEADS * LP * 3;
  const int choff = h * HD + grp * VC + c;

  // a wave owns 4 queries per trip and needs no barrier.  The trip's inputs (reference point, offset, logit of this lane's
  // sample; d_out of this lane's channel) are requested one trip ahead: the 16 waves of the block run the same phases at
  // about the same time, so a load issued where it is needed left the LDS atomic pipe idle for a whole memory round trip
  // per trip (round 2: 455 us per encoder layer against 145 us of atomics)
  const int l_own = j < LP ? j / P : 0;
  const int W = sel4(lv.W, l_own), H = sel4(lv.H, l_own), st = sel4(lv.start, l_own);
  const float Wf = (float)W, Hf = (float)H;
  const int g2 = (slot * VC) >> 4;                               // phase 2 lanes read the query of THEIR records
  struct In { float2 rr, oo; float lg, go; };
  auto fetch = [&](int qb) {
    In in;
    const int q = qb + g, q2 = qb + g2;
    const long long qrow = (long long)n * Lq + min(q, Lq - 1);
    const float* ow = offw + qrow * rowlen;
    const int jj = min(j, LP - 1);
    in.rr = *reinterpret_cast<const float2*>(ref + (qrow * L + l_own) * 2);
    in.oo = *reinterpret_cast<const float2*>(ow + (h * LP + jj) * 2);
    in.lg = ow[HEADS * LP * 2 + h * LP + jj];
    in.go = d_out[((long long)n * Lq + min(q2, Lq - 1)) * CH + choff];
    return in;
  };
  In cur = fetch(wv * 4);
  for (int qb = wv * 4; qb < Lq; qb += 64) {
    const In nxt = fetch(min(qb + 64, Lq - 1));                  // (clamped: the last trip's prefetch reads valid rows, unused)
    const int q = qb + g;
    const bool qlive = q < Lq;
    {
      const bool slive = qlive && j < LP;
      const float px = (cur.rr.x + cur.oo.x / Wf) * Wf - 0.5f;
      const float py = (cur.rr.y + cur.oo.y / Hf) * Hf - 0.5f;
      const float lg = slive ? cur.lg : -INFINITY;
      const float mx = max16(lg);
      const float e = slive ? __expf(lg - mx) : 0.f;
      const float sm = sum16(e);
      const float aw = slive ? e / sm : 0.f;
      const Tap t = make_tap(px, py, W, H, st, slive, S);
      rec_ids[wv][lane] = make_uint2(t.i00 | (t.i01 << 16), t.i10 | (t.i11 << 16));
      rec_w[wv][lane] = make_float4(aw * (1.f - t.fx) * (1.f - t.fy), aw * t.fx * (1.f - t.fy), aw * (1.f - t.fx) * t.fy,
                                    aw * t.fx * t.fy);
    }
    const float go = (qb + g2 < Lq) ? cur.go : 0.f;
    const int rbase = slot * VC;
#pragma unroll
    for (int s = 0; s < VC; ++s) {
      const uint2 ids = rec_ids[wv][rbase + s];
      const float4 w = rec_w[wv][rbase + s];
      atomicAdd(&slab[(ids.x & 0xFFFFu) * VS + c], (double)(go * w.x));
      atomicAdd(&slab[(ids.x >> 16) * VS + c], (double)(go * w.y));
      atomicAdd(&slab[(ids.y & 0xFFFFu) * VS + c], (double)(go * w.z));
      atomicAdd(&slab[(ids.y >> 16) * VS + c], (double)(go * w.w));
    }
    cur = nxt;
  }
  __syncthreads();
  float* dvb = d_value + (long long)n * S * CH + h * HD + grp * VC;
  if constexpr (VC >= 4) {
    constexpr int Q = VC / 4;
    for (int k = threadIdx.x; k < S * Q; k += 1024) {
      const int p_ = k / Q, c4 = (k % Q) * 4;
      const double* sp = &slab[p_ * VS + c4];
      *reinterpret_cast<float4*>(dvb + (long long)p_ * CH + c4) = make_float4((float)sp[0], (float)sp[1], (float)sp[2], (float)sp[3]);
    }
  } else {
    for (int k = threadIdx.x; k < S; k += 1024)
      *reinterpret_cast<float2*>(dvb + (long long)k * CH) = make_float2((float)slab[k * VS], (float)slab[k * VS + 1]);
  }
}

// Fixed-point form of the value kernel (the default of the bf16x3 product path; the exact-fp32 leg keeps the fp64 slab).
// An LDS accumulator is one 64-bit INTEGER holding TWO channels: ds_add_u64 costs 6.3 clocks per wave-instruction against 8.0
// for ds_add_f64 (tools/lab/lds_atomic_bench.hip) and covers two channels, so a block owns 16 channels of a head where the
// fp64 slab holds 8 -- half the blocks, and the sampling geometry of every query is rebuilt twice per head instead of 4 times.
//   * packing: a contribution pair (v0, v1), both int32, is added as the 64-bit value v1 * 2^32 + v0 (v0 sign-extended), so
//     the slab word is exactly (sum v1) * 2^32 + (sum v0): no carry error, decoded at the end as lo = (int32)word,
//     hi = (int32)(word >> 32) - (lo >> 31).
//   * scale: v = rint(go * w * 2^s) (one contribution <= 2^21) with s chosen per block from gmax = max |d_out| over the block's 16 channels and all
//     queries: the softmax weights of a (query, head) sum to 1 and the bilinear weights of a sample to <= 1, so one pixel
//     receives at most |go_q| per query and |sum| <= Lq * gmax; with 2^s = 2^30 / (pow2ceil(Lq) * pow2ceil(gmax)) a sum can
//     never leave int32 (rounding adds at most half a unit per contribution, < 2^17 units).  The quantum is
//     pow2ceil(Lq) * pow2ceil(gmax) * 2^-30 (Lq = 1360: <= 4e-6 gmax relative to the block's largest incoming gradient).
//   * a non-finite d_out (inf / NaN) has no fixed-point image: the block then writes NaN to its whole d_value slice.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VP>
__global__ void __launch_bounds__(1024) msda_bwd_value_fx_kernel(const float* __restrict__ d_out, const float* __restrict__ offw,
                                                                 const float* __restrict__ ref, float* __restrict__ d_value,
                                                                 Levels lv, int N, int S, int Lq, int L, int P) {
  constexpr int VC = 2 * VP, GROUPS = HD / VC, BPI = HEADS * GROUPS;   // channels per block, channel groups per head, blocks per image
  constexpr int SLOTS = 64 / VP;
  extern __shared__ __attribute__((aligned(16))) unsigned long long fslab[];   // [(S + 1)][VP]; row S swallows taps outside a level
  __shared__ uint4 rec_off[16][64];                              // per wave: [query g (4)][sample j (16)] -> slab byte offsets of the corners
  __shared__ float4 rec_w[16][64];
  __shared__ unsigned gmax_w[16];
  int n, sub;
  {
    const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
    n = (loc / BPI) * 8 + xcd;
    sub = loc % BPI;
    if (n >= N) return;
  }
  const int h = sub / GROUPS, grp = sub % GROUPS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ch0 = h * HD + grp * VC;
  for (int k = threadIdx.x; k < (S + 1) * VP; k += 1024) fslab[k] = 0ull;
  // block maximum of |d_out| over (all queries, the block's channels), as ordered bit patterns (NaN > inf > finite)
  unsigned gb = 0u;
  for (int k = threadIdx.x; k < Lq * VP; k += 1024) {
    const int q = k / VP, c2 = (k - q * VP) * 2;
    const float2 v = *reinterpret_cast<const float2*>(d_out + ((long long)n * Lq + q) * CH + ch0 + c2);
    gb = max(gb, max(__float_as_uint(fabsf(v.x)), __float_as_uint(fabsf(v.y))));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) gb = max(gb, (unsigned)__shfl_xor((int)gb, o));
  if (lane == 0) gmax_w[wv] = gb;
  __syncthreads();
  gb = gmax_w[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) gb = max(gb, gmax_w[k]);
  const bool finite = gb < 0x7f800000u;
  float scale = 0.f, inv_scale = 0.f;
  if (finite && gb != 0u) {
    int e;
    (void)frexpf(__uint_as_float(gb), &e);                       // gmax <= 2^e
    const int lq_bits = Lq > 1 ? 32 - __clz(Lq - 1) : 0;         // pow2ceil(Lq) = 2^lq_bits
    // one contribution stays below 2^21 (the float -> int conversion below is exact up to 2^22), a sum below 2^30
    const int sx = max(-126, min(126, min(30 - lq_bits, 21) - e));
    scale = ldexpf(1.f, sx);
    inv_scale = ldexpf(1.f, -sx);
  }

  const int g = lane >> 4, j = lane & 15;                        // phase 1: (query, sample)
  const int c = lane & (VP - 1), slot = lane / VP;               // phase 2: channel pair c; records slot*VP .. slot*VP+VP-1
  static_assert(SLOTS * VP == 64 && 16 % VP == 0, "a slot's records must stay inside one query");
  const int LP = L * P;
  const int rowlen = HEADS * LP * 3;
  const int choff = ch0 + 2 * c;
  const int l_own = j < LP ? j / P : 0;
  const int W = sel4(lv.W, l_own), H = sel4(lv.H, l_own), st = sel4(lv.start, l_own);
  const float Wf = (float)W, Hf = (float)H;
  const int g2 = (slot * VP) >> 4;
  struct In { float2 rr, oo, go; float lg; };
  auto fetch = [&](int qb) {
    In in;
    const int q = qb + g, q2 = qb + g2;
    const long long qrow = (long long)n * Lq + min(q, Lq - 1);
    const float* ow = offw + qrow * rowlen;
    const int jj = min(j, LP - 1);
    in.rr = *reinterpret_cast<const float2*>(ref + (qrow * L + l_own) * 2);
    in.oo = *reinterpret_cast<const float2*>(ow + (h * LP + jj) * 2);
    in.lg = ow[HEADS * LP * 2 + h * LP + jj];
    in.go = *reinterpret_cast<const float2*>(d_out + ((long long)n * Lq + min(q2, Lq - 1)) * CH + choff);
    return in;
  };
  // rint(g * w) for both channels of the pair without a convert: one packed FMA onto 1.5 * 2^23 leaves the rounded integer
  // (round-to-nearest-even of the exact product, |.| <= 2^22) in the low mantissa bits; the pair then goes out as
  // hi * 2^32 + lo with lo sign-extended (see the kernel header)
  char* const slab_c = reinterpret_cast<char*>(fslab) + c * 8;
  auto add2 = [&](unsigned off, float wgt, v2f gg) {
    const v2f t = __builtin_elementwise_fma(gg, (v2f)(wgt), (v2f)(12582912.f));
    const int v0 = (int)__float_as_uint(t.x) - 0x4B400000, v1 = (int)__float_as_uint(t.y) - 0x4B400000;
    const unsigned long long pk = ((unsigned long long)(unsigned)(v1 + (v0 >> 31)) << 32) | (unsigned)v0;
    atomicAdd(reinterpret_cast<unsigned long long*>(slab_c + off), pk);
  };
  In cur = fetch(wv * 4);
  for (int qb = wv * 4; qb < Lq; qb += 64) {
    const In nxt = fetch(min(qb + 64, Lq - 1));
    const int q = qb + g;
    const bool qlive = q < Lq;
    {
      const bool slive = qlive && j < LP;
      const float px = (cur.rr.x + cur.oo.x / Wf) * Wf - 0.5f;
      const float py = (cur.rr.y + cur.oo.y / Hf) * Hf - 0.5f;
      const float lg = slive ? cur.lg : -INFINITY;
      const float mx = max16(lg);
      const float e = slive ? __expf(lg - mx) : 0.f;
      const float sm = sum16(e);
      const float aw = slive ? e / sm : 0.f;
      const Tap t = make_tap(px, py, W, H, st, slive, S);
      rec_off[wv][lane] = make_uint4(t.i00 * (VP * 8), t.i01 * (VP * 8), t.i10 * (VP * 8), t.i11 * (VP * 8));
      rec_w[wv][lane] = make_float4(aw * (1.f - t.fx) * (1.f - t.fy), aw * t.fx * (1.f - t.fy), aw * (1.f - t.fx) * t.fy,
                                    aw * t.fx * t.fy);
    }
    const bool live2 = qb + g2 < Lq;
    v2f gg;
    gg.x = live2 ? cur.go.x * scale : 0.f;
    gg.y = live2 ? cur.go.y * scale : 0.f;
    const int rbase = slot * VP;
#pragma unroll
    for (int s = 0; s < VP; ++s) {
      const uint4 off = rec_off[wv][rbase + s];
      const float4 w = rec_w[wv][rbase + s];
      add2(off.x, w.x, gg);
      add2(off.y, w.y, gg);
      add2(off.z, w.z, gg);
      add2(off.w, w.w, gg);
    }
    cur = nxt;
  }
  __syncthreads();
  float* dvb = d_value + (long long)n * S * CH + ch0;
  const float nanv = __uint_as_float(0x7fc00000u);
  auto dec = [&](unsigned long long wd, float& lo, float& hi) {
    const int l = (int)(unsigned)wd;
    const int hh = (int)(unsigned)(wd >> 32) - (l >> 31);
    lo = finite ? (float)l * inv_scale : nanv;
    hi = finite ? (float)hh * inv_scale : nanv;
  };
  constexpr int Q = VP / 2;                                      // float4 stores per pixel row
  for (int k = threadIdx.x; k < S * Q; k += 1024) {
    const int p_ = k / Q, c2 = (k % Q) * 2;
    float4 o;
    dec(fslab[p_ * VP + c2], o.x, o.y);
    dec(fslab[p_ * VP + c2 + 1], o.z, o.w);
    *reinterpret_cast<float4*>(dvb + (long long)p_ * CH + c2 * 2) = o;
  }
}

int fill_levels(Levels& lv, const int* shapes, const int* level_start, int L, int S) {
  CAPE_REQUIRE(L >= 1 && L <= 4, "cape_msda: L=%d must be in 1..4", L);
  long long tot = 0;
  for (int l = 0; l < 4; ++l) {
    lv.H[l] = l < L ? shapes[2 * l] : 1;
    lv.W[l] = l < L ? shapes[2 * l + 1] : 1;
    lv.start[l] = l < L ? level_start[l] : 0;
    if (l < L) {
      CAPE_REQUIRE(lv.H[l] > 0 && lv.W[l] > 0 && lv.start[l] == tot, "cape_msda: level %d shape/start inconsistent", l);
      tot += (long long)lv.H[l] * lv.W[l];
    }
  }
  CAPE_REQUIRE(tot == S, "cape_msda: sum(H*W)=%lld != S=%d", tot, S);
  return 0;
}

}  // namespace

extern "C" int cape_msda_fwd(const float* value, const float* offw, const float* ref, const int* shapes,
                             const int* level_start, float* out, int N, int S, int Lq, int L, int P,
                             cape_stream_t stream) {
  CAPE_REQUIRE(value && offw && ref && shapes && level_start && out, "cape_msda_fwd: null pointer");
  CAPE_REQUIRE(P >= 1 && L * P <= 16, "cape_msda_fwd: L*P=%d must be <= 16", L * P);
  if (N <= 0 || Lq <= 0) return 0;
  Levels lv;
  if (fill_levels(lv, shapes, level_start, L, S)) return 1;
  const int bpi = (Lq + 3) / 4;
  const int imgs_per_x = (N + 7) / 8;
  const long long blocks = (long long)imgs_per_x * bpi * 8;
  CAPE_REQUIRE(blocks < (1ll << 31), "cape_msda_fwd: grid too large");
  static const bool shuffle_form = getenv("CAPE_MSDA_FWD_SHUFFLE") != nullptr;      // tuning switch
  if (S < 65535 && !shuffle_form)
    hipLaunchKernelGGL(msda_fwd_rec_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), value, offw, ref, out, lv,
                       N, S, Lq, L, P, bpi);
  else
    hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), value, offw, ref, out, lv,
                       N, S, Lq, L, P, bpi);
  CAPE_LAUNCH_CHECK("cape_msda_fwd");
  return 0;
}

static int msda_bwd_check(const float* d_out, const float* value, const float* offw, const float* ref, const int* shapes,
                          const int* level_start, float* d_value, float* d_offw, int L, int P) {
  CAPE_REQUIRE(d_out && value && offw && ref && shapes && level_start && d_value && d_offw, "cape_msda_bwd: null pointer");
  CAPE_REQUIRE(P >= 1 && L * P <= 16, "cape_msda_bwd: L*P=%d must be <= 16", L * P);
  return 0;
}

extern "C" int cape_msda_bwd_atomic(const float* d_out, const float* value, const float* offw, const float* ref,
                                    const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                                    int N, int S, int Lq, int L, int P, cape_stream_t stream) {
  if (msda_bwd_check(d_out, value, offw, ref, shapes, level_start, d_value, d_offw, L, P)) return 1;
  if (N <= 0 || Lq <= 0) return 0;
  Levels lv;
  if (fill_levels(lv, shapes, level_start, L, S)) return 1;
  hipError_t e = hipMemsetAsync(d_value, 0, sizeof(float) * (size_t)N * S * CH, as_stream(stream));
  if (e == hipSuccess && d_ref) e = hipMemsetAsync(d_ref, 0, sizeof(float) * (size_t)N * Lq * L * 2, as_stream(stream));
  if (e != hipSuccess) return cape_set_error("cape_msda_bwd_atomic: memset: %s", hipGetErrorString(e));
  const long long waves = (long long)N * Lq * 4;
  const long long blocks = (waves + 3) / 4;
  CAPE_REQUIRE(blocks < (1ll << 31), "cape_msda_bwd_atomic: grid too large");
  hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_out, value, offw, ref,
                     d_value, d_offw, d_ref, lv, N, S, Lq, L, P, waves);
  CAPE_LAUNCH_CHECK("cape_msda_bwd_atomic");
  return 0;
}

extern "C" int cape_msda_bwd(const float* d_out, const float* value, const float* offw, const float* ref,
                             const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                             int N, int S, int Lq, int L, int P, cape_stream_t stream) {
  return cape_msda_bwd_ex(d_out, value, offw, ref, shapes, level_start, d_value, d_offw, d_ref, N, S, Lq, L, P, 0, stream);
}

extern "C" int cape_msda_bwd_ex(const float* d_out, const float* value, const float* offw, const float* ref,
                                const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                                int N, int S, int Lq, int L, int P, int value_accum, cape_stream_t stream) {
  CAPE_REQUIRE(value_accum == 0 || value_accum == 1, "cape_msda_bwd_ex: value_accum must be 0 (fp64 slab) or 1 (fixed-point pairs)");
  if (msda_bwd_check(d_out, value, offw, ref, shapes, level_start, d_value, d_offw, L, P)) return 1;
  if (N <= 0 || Lq <= 0) return 0;
  // split form when the (image, head, 8-channel) fp64 slab fits in LDS (S <= ~2300: every image size up to 320x320);
  // larger geometries take the memory-side-atomic form
  const size_t kMaxSlab = 160 * 1024 - 16 * 64 * (sizeof(uint4) + sizeof(float4)) - 512;
  int vc = 8;                                                    // widest channel group whose slab fits
  while (vc >= 2 && (size_t)(S + 1) * vc * sizeof(double) > kMaxSlab) vc >>= 1;
  const size_t slab_bytes = (size_t)(S + 1) * vc * sizeof(double);
  static const bool force_atomic = getenv("CAPE_MSDA_BWD_ATOMIC") != nullptr;      // tuning switch
  const long long imgs8 = ((long long)N + 7) / 8;
  const int bpi = (Lq + 3) / 4;
  const long long vblocks = imgs8 * 8 * HEADS * (HD / (vc < 2 ? 2 : vc));
  if (force_atomic || vc < 2 || S >= 65535 || vblocks >= (1ll << 31) || imgs8 * 8 * bpi >= (1ll << 31))
    return cape_msda_bwd_atomic(d_out, value, offw, ref, shapes, level_start, d_value, d_offw, d_ref, N, S, Lq, L, P, stream);
  Levels lv;
  if (fill_levels(lv, shapes, level_start, L, S)) return 1;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_kernel<8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_kernel<4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_kernel<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_fx_kernel<8>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_fx_kernel<4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_fx_kernel<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxSlab);
    if (e != hipSuccess) return cape_set_error("cape_msda_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (d_ref && P != 4) {                                         // (P == 4: the gather kernel writes every d_ref element itself)
    hipError_t e = hipMemsetAsync(d_ref, 0, sizeof(float) * (size_t)N * Lq * L * 2, as_stream(stream));
    if (e != hipSuccess) return cape_set_error("cape_msda_bwd: memset: %s", hipGetErrorString(e));
  }
#define VALUE_LAUNCH(W)                                                                                                     \
  hipLaunchKernelGGL((msda_bwd_value_kernel<W>), dim3((unsigned)vblocks), dim3(1024), slab_bytes, as_stream(stream), d_out, offw, \
                     ref, d_value, lv, N, S, Lq, L, P)
#define VALUE_LAUNCH_FX(W)                                                                                                  \
  hipLaunchKernelGGL((msda_bwd_value_fx_kernel<W>), dim3((unsigned)fxblocks), dim3(1024), fx_bytes, as_stream(stream),           \
                     d_out, offw, ref, d_value, lv, N, S, Lq, L, P)
  if (value_accum == 1) {                                        // same slab bytes (8 per accumulator), twice the channels per block
    int vp = vc;                                                 // few images: narrower blocks until every CU has one
    while (vp > 2 && imgs8 * 8 * HEADS * (HD / (2 * vp)) < 256) vp >>= 1;
    const long long fxblocks = imgs8 * 8 * HEADS * (HD / (2 * vp));
    const size_t fx_bytes = (size_t)(S + 1) * vp * sizeof(unsigned long long);
    if (vp == 8) VALUE_LAUNCH_FX(8); else if (vp == 4) VALUE_LAUNCH_FX(4); else VALUE_LAUNCH_FX(2);
  } else {
    if (vc == 8) VALUE_LAUNCH(8); else if (vc == 4) VALUE_LAUNCH(4); else VALUE_LAUNCH(2);
  }
#undef VALUE_LAUNCH_FX
#undef VALUE_LAUNCH
  hipLaunchKernelGGL(msda_bwd_offw_kernel, dim3((unsigned)(imgs8 * 8 * bpi)), dim3(256), 0, as_stream(stream), d_out, value, offw,
                     ref, d_offw, d_ref, lv, N, S, Lq, L, P, bpi);
  CAPE_LAUNCH_CHECK("cape_msda_bwd");
  return 0;
}
