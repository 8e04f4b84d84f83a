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
// backward, LDS-accumulating form.  One block per (image, head, level group): the d_value slab of that
// (image, head) for the group's levels lives in LDS (level 0: H0*W0*128 B; remaining levels together),
// every tap is an LDS atomic (ds_add_f32) instead of a memory-side atomic, and the slab is written back
// with plain stores (exclusive owner -> d_value needs no zero fill).  32 query slots x 32 channels per block
// (16 waves: the loop is a chain of dependent L2 gathers, so memory-level parallelism per CU is what counts).
// d_aw is emitted raw (gradient w.r.t. the softmaxed weight); msda_softmax_bwd_kernel turns it into the
// logit gradient afterwards (the softmax spans samples handled by both groups).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) msda_bwd_lds_kernel(const float* __restrict__ d_out, const float* __restrict__ value,
                                                            const float* __restrict__ offw, const float* __restrict__ ref,
                                                            float* __restrict__ d_value, float* __restrict__ d_offw,
                                                            float* __restrict__ d_ref, Levels lv, int N, int S, int Lq, int L,
                                                            int P, int lvl_begin, int lvl_end) {
  extern __shared__ __attribute__((aligned(16))) float dval[];     // [npix][32]
  const int n = blockIdx.x >> 3, h = blockIdx.x & 7;
  const int pix0 = sel4(lv.start, lvl_begin);
  const int pix1 = (lvl_end < L) ? sel4(lv.start, lvl_end) : S;
  const int npix = pix1 - pix0;
  for (int i = threadIdx.x; i < npix * HD; i += 1024) dval[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int c = lane & 31;
  const int slot = threadIdx.x >> 5;                               // 32 query slots
  const int LP = L * P;
  const int j = c & 15;
  const int rowlen = HEADS * LP * 3;
  const float* vbase = value + (long long)n * S * CH + h * HD + c;
  const int jb = lvl_begin * P, je = lvl_end * P;                  // this group's sample range
  const int iters = (Lq + 31) >> 5;
  for (int it = 0; it < iters; ++it) {
    const int q = it * 32 + slot;
    const bool qlive = q < Lq;                                     // half-wave uniform
    const long long qrow = (long long)n * Lq + (qlive ? q : 0);
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
    const float aw = e / sm;
    const float go = qlive ? d_out[qrow * CH + h * HD + c] : 0.f;
    float my_daw = 0.f, my_dpx = 0.f, my_dpy = 0.f;
    // samples in chunks of 4: all 16 gathers of a chunk are issued before any is consumed (the loop is a chain of
    // dependent L2 gathers; memory-level parallelism per wave is what bounds it)
    for (int jc = jb; jc < je; jc += 4) {
      float v[4][4], fxs[4], fys[4], gas[4];
      int base[4], Ws[4], bits[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int jj = jc + s_;
        const int src = (lane & 32) | (jj & 15);
        const float x = __shfl(px, src, 64), y = __shfl(py, src, 64), a = __shfl(aw, src, 64);
        const int l = __shfl(l_own, src, 64);
        const int W = sel4(lv.W, l), H = sel4(lv.H, l);
        const float xf = floorf(x), yf = floorf(y);
        fxs[s_] = x - xf; fys[s_] = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        const bool ok = qlive && jj < je;
        const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
        const bool ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
        bits[s_] = ok ? ((ya && xa) | ((ya && xb) << 1) | ((yb && xa) << 2) | ((yb && xb) << 3)) : 0;
        base[s_] = sel4(lv.start, l) + y0 * W + x0;
        Ws[s_] = W;
        gas[s_] = go * a;
        v[s_][0] = (bits[s_] & 1) ? vbase[(long long)base[s_] * CH] : 0.f;
        v[s_][1] = (bits[s_] & 2) ? vbase[(long long)(base[s_] + 1) * CH] : 0.f;
        v[s_][2] = (bits[s_] & 4) ? vbase[(long long)(base[s_] + W) * CH] : 0.f;
        v[s_][3] = (bits[s_] & 8) ? vbase[(long long)(base[s_] + W + 1) * CH] : 0.f;
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int jj = jc + s_;
        const float fx = fxs[s_], fy = fys[s_], ga = gas[s_];
        const int b_ = base[s_] - pix0;
        if (bits[s_] & 1) atomicAdd(&dval[b_ * HD + c], ga * (1.f - fx) * (1.f - fy));
        if (bits[s_] & 2) atomicAdd(&dval[(b_ + 1) * HD + c], ga * fx * (1.f - fy));
        if (bits[s_] & 4) atomicAdd(&dval[(b_ + Ws[s_]) * HD + c], ga * (1.f - fx) * fy);
        if (bits[s_] & 8) atomicAdd(&dval[(b_ + Ws[s_] + 1) * HD + c], ga * fx * fy);
        const float v00 = v[s_][0], v01 = v[s_][1], v10 = v[s_][2], v11 = v[s_][3];
        const float samp = (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
        const float dsx = (1.f - fy) * (v01 - v00) + fy * (v11 - v10);
        const float dsy = (1.f - fx) * (v10 - v00) + fx * (v11 - v01);
        const float r_aw = half_sum32(go * samp);
        const float r_px = half_sum32(ga * dsx);
        const float r_py = half_sum32(ga * dsy);
        if (j == jj && jj < je) { my_daw = r_aw; my_dpx = r_px; my_dpy = r_py; }
      }
    }
    const bool own = qlive && c < 16 && j >= jb && j < je;
    float* dow = d_offw + qrow * rowlen;
    if (own) {
      dow[(h * LP + j) * 2 + 0] = my_dpx;
      dow[(h * LP + j) * 2 + 1] = my_dpy;
      dow[HEADS * LP * 2 + h * LP + j] = my_daw;                   // raw d_aw; softmax backward runs afterwards
    }
    if (d_ref) {
      float rx = own ? my_dpx * (float)sel4(lv.W, l_own) : 0.f;
      float ry = own ? my_dpy * (float)sel4(lv.H, l_own) : 0.f;
      if (P == 4) {
        rx += __shfl_xor(rx, 1, 64); rx += __shfl_xor(rx, 2, 64);
        ry += __shfl_xor(ry, 1, 64); ry += __shfl_xor(ry, 2, 64);
        if (own && (j & 3) == 0) {
          atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 0], rx);
          atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 1], ry);
        }
      } else if (own) {
        atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 0], rx);
        atomicAdd(&d_ref[(qrow * L + l_own) * 2 + 1], ry);
      }
    }
  }
  __syncthreads();
  float* dvb = d_value + ((long long)n * S + pix0) * CH + h * HD;
  for (int i = threadIdx.x; i < npix * 8; i += 1024) {
    const int p_ = i >> 3, c4 = (i & 7) * 4;
    *reinterpret_cast<float4*>(dvb + (long long)p_ * CH + c4) = *reinterpret_cast<const float4*>(&dval[p_ * HD + c4]);
  }
}

// dlogit = aw * (d_aw - sum_k aw_k d_aw_k) in place on the logit slots of d_offw; thread = (query, head)
__global__ void msda_softmax_bwd_kernel(const float* __restrict__ offw, float* __restrict__ d_offw, long long rows, int LP) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * HEADS) return;
  const long long qrow = i >> 3;
  const int h = (int)(i & 7);
  const int rowlen = HEADS * LP * 3;
  const float* lg = offw + qrow * rowlen + HEADS * LP * 2 + h * LP;
  float* dg = d_offw + qrow * rowlen + HEADS * LP * 2 + h * LP;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 16; ++k) if (k < LP) mx = fmaxf(mx, lg[k]);
  float e[16], d[16], sm = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { e[k] = k < LP ? __expf(lg[k] - mx) : 0.f; d[k] = k < LP ? dg[k] : 0.f; sm += e[k]; }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) { e[k] /= sm; dot += e[k] * d[k]; }
#pragma unroll
  for (int k = 0; k < 16; ++k) if (k < LP) dg[k] = e[k] * (d[k] - dot);
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
  hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), value, offw, ref, out, lv,
                     N, S, Lq, L, P, bpi);
  CAPE_LAUNCH_CHECK("cape_msda_fwd");
  return 0;
}

extern "C" int cape_msda_bwd(const float* d_out, const float* value, const float* offw, const float* ref,
                             const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                             int N, int S, int Lq, int L, int P, cape_stream_t stream) {
  CAPE_REQUIRE(d_out && value && offw && ref && shapes && level_start && d_value && d_offw, "cape_msda_bwd: null pointer");
  CAPE_REQUIRE(P >= 1 && L * P <= 16, "cape_msda_bwd: L*P=%d must be <= 16", L * P);
  if (N <= 0 || Lq <= 0) return 0;
  Levels lv;
  if (fill_levels(lv, shapes, level_start, L, S)) return 1;
  // LDS-accumulating form when the (image, head) gradient slabs fit: group A = level 0, group B = the rest
  const long long pixA = (long long)lv.H[0] * lv.W[0], pixB = (long long)S - pixA;
  const size_t ldsA = (size_t)pixA * HD * sizeof(float), ldsB = (size_t)pixB * HD * sizeof(float);
  const size_t kMaxLds = 144 * 1024;
  // Measured on MI355X (round 1, N=32, S=Lq=1360): the LDS form is SLOWER than memory-side atomics (2.0 ms vs
  // 1.7 ms per encoder layer): ds_add_f32 throughput per CU is no better than the chip-wide global float-atomic
  // rate per CU, and one (image, head) per CU leaves too few waves.  Kept opt-in for further tuning.
  static const bool use_lds = getenv("CAPE_MSDA_BWD_LDS") != nullptr;
  if (use_lds && L >= 2 && ldsA <= kMaxLds && ldsB <= kMaxLds && (long long)N * HEADS < (1ll << 31)) {
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
      if (e != hipSuccess) return cape_set_error("cape_msda_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      attr_done = true;
    }
    if (d_ref) {
      hipError_t e = hipMemsetAsync(d_ref, 0, sizeof(float) * (size_t)N * Lq * L * 2, as_stream(stream));
      if (e != hipSuccess) return cape_set_error("cape_msda_bwd: memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(msda_bwd_lds_kernel, dim3((unsigned)(N * HEADS)), dim3(1024), ldsA, as_stream(stream), d_out, value, offw,
                       ref, d_value, d_offw, d_ref, lv, N, S, Lq, L, P, 0, 1);
    hipLaunchKernelGGL(msda_bwd_lds_kernel, dim3((unsigned)(N * HEADS)), dim3(1024), ldsB, as_stream(stream), d_out, value, offw,
                       ref, d_value, d_offw, d_ref, lv, N, S, Lq, L, P, 1, L);
    const long long rows = (long long)N * Lq;
    hipLaunchKernelGGL(msda_softmax_bwd_kernel, dim3((unsigned)((rows * HEADS + 255) / 256)), dim3(256), 0, as_stream(stream),
                       offw, d_offw, rows, L * P);
    CAPE_LAUNCH_CHECK("cape_msda_bwd(lds)");
    return 0;
  }
  // fallback: memory-side atomics straight into d_value (any geometry); d_value / d_ref are zeroed here
  {
    hipError_t e = hipMemsetAsync(d_value, 0, sizeof(float) * (size_t)N * S * CH, as_stream(stream));
    if (e == hipSuccess && d_ref) e = hipMemsetAsync(d_ref, 0, sizeof(float) * (size_t)N * Lq * L * 2, as_stream(stream));
    if (e != hipSuccess) return cape_set_error("cape_msda_bwd: memset: %s", hipGetErrorString(e));
  }
  const long long waves = (long long)N * Lq * 4;
  const long long blocks = (waves + 3) / 4;
  CAPE_REQUIRE(blocks < (1ll << 31), "cape_msda_bwd: grid too large");
  hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_out, value, offw, ref,
                     d_value, d_offw, d_ref, lv, N, S, Lq, L, P, waves);
  CAPE_LAUNCH_CHECK("cape_msda_bwd");
  return 0;
}
