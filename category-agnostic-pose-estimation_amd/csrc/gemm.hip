// gemm.hip -- implicit-GEMM family on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// One kernel template covers nn.Linear fwd/dgrad/wgrad and NHWC convolution fwd/dgrad/wgrad
// (the A/B operand "modes" of cape_hip.h).  Design (MI355X):
//   * 256 threads = 4 wave64 in a 2x2 arrangement; block tile 64x64 (128x128 kept for tuning), BK = 32;
//   * operands are staged global -> registers -> LDS with branch-free 16-byte loads from clamped addresses;
//     two LDS buffers: tile t+1 is written in the middle of tile t's MFMAs, one barrier per k-tile;
//   * K-contiguous sources are kept row-major in LDS with a 36-float row stride (conflict-free
//     ds_read_b128: every lane fetches 4 consecutive k of its row); the physical k order inside a
//     group of 8 is permuted identically for A and B so one b128 read feeds 4 MFMAs
//     (MFMA step s of group g multiplies k = 8g + 4*(lane>>5) + s);
//   * M/N-contiguous sources (transposed operands, gathers along channels) are kept [k][mn] and
//     read with ds_read_b32 (32 consecutive floats per half wave: conflict-free);
//   * fp32 MFMA is 64 cycles per 32x32x2 step, so LDS and issue bandwidth are far from binding;
//     what matters is grid fill (>= 2 tiles per CU or split-K) and L2 locality (XCD-aware tile order).
#include <stdlib.h>
#include <type_traits>
#include "gemm_common.h"

namespace {

// guarded 4-float load: `valid` leading elements exist (0..4); vector path needs 16-B alignment
__device__ __forceinline__ float4 ldg4(const float* p, int valid) {
  if (valid >= 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const float4*>(p);
  float4 r = zero4();
  if (valid > 0) r.x = p[0];
  if (valid > 1) r.y = p[1];
  if (valid > 2) r.z = p[2];
  if (valid > 3) r.w = p[3];
  return r;
}

// KFULL (host-checked: K % 32 == 0): no k-tail handling at all in the loads
template <int BM, int BN, int AMODE, int BMODE, bool VEC, int PREC, bool KFULL>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmP pin) {
  GemmP p = pin;
  if (gridDim.y > 1) {                                             // batched launch: uniform operand offsets
    const int b0 = blockIdx.y / p.bdiv, b1 = blockIdx.y - b0 * p.bdiv;
    p.A += b0 * p.sA0 + b1 * p.sA1;
    p.B += b0 * p.sB0 + b1 * p.sB1;
    p.C += b0 * p.sC0 + b1 * p.sC1;
  }
  constexpr bool A_KC = (AMODE == 0 || AMODE == 2 || AMODE == 3);
  constexpr bool B_KC = (BMODE == 0);
  constexpr int A_LD = A_KC ? 36 : (BM + 4);
  constexpr int B_LD = B_KC ? 36 : (BN + 4);
  constexpr int A_SZ = A_KC ? BM * 36 : BK * (BM + 4);
  constexpr int B_SZ = B_KC ? BN * 36 : BK * (BN + 4);
  constexpr int NA = BM / 32;   // 16-byte chunks per thread per k-tile
  constexpr int NB = BN / 32;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TI = WTM / 32, TJ = WTN / 32;

  // two LDS buffers per operand: tile t+1 is written into the other buffer in the middle of tile t's MFMAs
  // (one barrier per k-tile; a single wave per SIMD keeps the matrix pipe fed)
  // PREC 1: per operand and buffer two bf16 planes (hi, lo) laid out [row][k] with an 80-byte row stride
  // (32 k x 2 B + 16 B pad: conflict-free ds_read_b128 of 8 consecutive k per lane)
  constexpr int PL_LD = 40;                                    // bf16 elements per row ([row][k] image)
  constexpr int PMA_LD = BM + 4, PMB_LD = BN + 4;              // 32-bit words per k-pair row ([k/2][mn] image)
  constexpr int A_PL = BM * PL_LD, B_PL = BN * PL_LD;          // elements per plane
  constexpr int A_WORDS = PREC ? A_PL : A_SZ;                  // 2 planes x A_PL bf16 = A_PL 32-bit words
  constexpr int B_WORDS = PREC ? B_PL : B_SZ;
  __shared__ __attribute__((aligned(16))) float As[2][A_WORDS];
  __shared__ __attribute__((aligned(16))) float Bs[2][B_WORDS];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware work order.  Blocks b and b+8 share an XCD (and its L2).  The grid is 1-D over
  //      (k-splits x tiles).  Without split-K each XCD gets a contiguous run of tile ids (neighbouring tiles
  //      share A rows / B columns).  With split-K all tiles of one k-split read the same k-slab of both operands,
  //      so a split is pinned to one XCD (split = xcd + 8*i): the slab is fetched from HBM once per XCD instead of
  //      once per tile (measured: the 256x256x43520 wgrad was fabric-bound, 8x read amplification, before this).
  const int ntiles = p.tilesM * p.tilesN;
  int tile, split;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    if (p.split_k > 1 && (p.split_k & 7) == 0) {
      split = xcd + 8 * (loc / ntiles);
      tile = loc % ntiles;
    } else if (p.split_k > 1) {
      split = bid / ntiles;
      tile = bid - split * ntiles;
    } else {
      const int q = ntiles >> 3, r = ntiles & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
      split = 0;
    }
  }
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- split-K range
  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int kt_begin = split * per;
  const int kt_end = min(ktiles, kt_begin + per);
  if (kt_begin >= kt_end) return;

  // ---- per-thread load coordinates
  // K-contig: chunk column kc = t&7 (k offset 4kc), rows (t>>3) + 32j
  // MN-contig: chunks per k-row CH = B?/4; column mc = t % CH, k-row (t / CH) + (256/CH) j
  // PREC 1 stores 8 bytes per lane per plane with an 80-byte row stride: a 16-lane store group covers two rows, which
  // must sit 4 rows apart (4 x 80 B = 16 banks) to be conflict-free, so the 8-lane row groups are dealt 0,4,1,5,2,6,3,7
  const int rg = t >> 3;
  const int rperm = PREC ? ((rg & 24) | ((rg & 1) << 2) | ((rg >> 1) & 3)) : rg;
  const int a_kc = t & 7, a_r0 = rperm;
  constexpr int A_CH = BM / 4;
  const int a_mc = t % A_CH, a_k0 = t / A_CH;
  constexpr int A_KSTEP = 256 / A_CH;
  const int b_kc = t & 7, b_r0 = rperm;
  constexpr int B_CH = BN / 4;
  const int b_mc = t % B_CH, b_k0 = t / B_CH;
  constexpr int B_KSTEP = 256 / B_CH;

  // conv gather row decode (AMODE 2: rows are output positions; AMODE 3: rows are input positions)
  int a_n[NA], a_y[NA], a_x[NA];
  if constexpr (AMODE == 2 || AMODE == 3) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int row = m0 + a_r0 + 32 * j;
      if (row < p.M) {
        const int RW = (AMODE == 2) ? p.cOW : p.cW;
        const int RH = (AMODE == 2) ? p.cOH : p.cH;
        const int x = row % RW;
        const int tq = row / RW;
        const int y = tq % RH;
        a_n[j] = tq / RH;
        if (AMODE == 2) { a_y[j] = y * p.cStride - p.cPad; a_x[j] = x * p.cStride - p.cPad; }
        else { a_y[j] = y + p.cPad; a_x[j] = x + p.cPad; }
      } else {
        a_n[j] = -1; a_y[j] = 0; a_x[j] = 0;
      }
    }
  }

  // register stages: DEPTH k-tiles of both operands in flight between global memory and the LDS store.  Measured on
  // MI355X: DEPTH 3 (64x64) / 2 (128x128) is 5-10 % SLOWER than 1 on every shape of tools/gemm_bench.py -- the loop is
  // bound by instruction issue (VALU split + LDS traffic), not by memory latency, and the extra live registers cost more
  // than the latency they hide.  The deeper pipeline is kept for tuning.
  constexpr int DEPTH = 1;
  float4 ra[DEPTH][NA], rb[DEPTH][NB];

  // VEC (host-checked: 16-byte aligned bases, leading dimensions and contiguous extents multiples of 4):
  // every load is an unconditional 16-byte load from a CLAMPED (always valid) address; rows beyond M/N only feed
  // accumulators that the epilogue never stores, chunks beyond K are zeroed with a select -- no branches, no
  // scalar loads in the main loop.  !VEC keeps the guarded element-wise path for odd shapes (K = 2, ld = 3, ...).
  auto load_tiles = [&](int kt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int kbase = kt * BK;
    // ---------------- A ----------------
    if constexpr (AMODE == 0) {
      const int k = kbase + 4 * a_kc;
      if constexpr (VEC) {
        const bool kin = KFULL || k < p.K;
        const int kc_ = kin ? k : 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int row = min(m0 + a_r0 + 32 * j, p.M - 1);
          const float4 v = *reinterpret_cast<const float4*>(p.A + (long long)row * p.lda + kc_);
          ra[sl][j] = kin ? v : zero4();
        }
      } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int row = m0 + a_r0 + 32 * j;
          const int valid = (row < p.M) ? min(4, max(0, p.K - k)) : 0;
          ra[sl][j] = valid ? ldg4(p.A + (long long)row * p.lda + k, valid) : zero4();
        }
      }
    } else if constexpr (AMODE == 1) {
      const int mcol = m0 + 4 * a_mc;
      if constexpr (VEC) {
        const int mc_ = min(mcol, p.M - 4);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int k = kbase + (PREC ? 2 * (a_k0 + A_KSTEP * (j >> 1)) + (j & 1) : a_k0 + A_KSTEP * j);
          const bool kin = KFULL || k < p.K;
          const float4 v = *reinterpret_cast<const float4*>(p.A + (long long)(kin ? k : 0) * p.lda + mc_);
          ra[sl][j] = kin ? v : zero4();
        }
      } else {
        const int vm = min(4, max(0, p.M - mcol));
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int k = kbase + (PREC ? 2 * (a_k0 + A_KSTEP * (j >> 1)) + (j & 1) : a_k0 + A_KSTEP * j);
          ra[sl][j] = (k < p.K && vm) ? ldg4(p.A + (long long)k * p.lda + mcol, vm) : zero4();
        }
      }
    } else if constexpr (AMODE == 2) {
      const int k = kbase + 4 * a_kc;
      const int kq = (KFULL || k < p.K) ? k : 0;
      const int tap = kq / p.cC, c = kq - tap * p.cC;
      const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int iy = a_y[j] + kh, ix = a_x[j] + kw;
        const bool ok = (a_n[j] >= 0) && (KFULL || k < p.K) && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
        const int nn = max(a_n[j], 0), yy = min(max(iy, 0), p.cH - 1), xx = min(max(ix, 0), p.cW - 1);
        const float4 v = *reinterpret_cast<const float4*>(p.A + (((long long)nn * p.cH + yy) * p.cW + xx) * p.cC + c);
        ra[sl][j] = ok ? v : zero4();
      }
    } else {  // AMODE == 3: dgrad gather of dY (N, OH, OW, O); k = tap*O + o
      const int k = kbase + 4 * a_kc;
      const int kq = (KFULL || k < p.K) ? k : 0;
      const int tap = kq / p.cO, o = kq - tap * p.cO;
      const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int ty = a_y[j] - kh, tx = a_x[j] - kw;
        bool ok = (a_n[j] >= 0) && (KFULL || k < p.K) && ty >= 0 && tx >= 0;
        int oy = ty, ox = tx;
        if (p.cStride != 1) {
          oy = ty / p.cStride; ox = tx / p.cStride;
          ok = ok && (oy * p.cStride == ty) && (ox * p.cStride == tx);
        }
        ok = ok && oy < p.cOH && ox < p.cOW;
        const int nn = max(a_n[j], 0), yy = min(max(oy, 0), p.cOH - 1), xx = min(max(ox, 0), p.cOW - 1);
        const float4 v = *reinterpret_cast<const float4*>(p.A + (((long long)nn * p.cOH + yy) * p.cOW + xx) * p.cO + o);
        ra[sl][j] = ok ? v : zero4();
      }
    }
    // ---------------- B ----------------
    if constexpr (BMODE == 0) {
      const int k = kbase + 4 * b_kc;
      if constexpr (VEC) {
        const bool kin = KFULL || k < p.K;
        const int kc_ = kin ? k : 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int row = min(n0 + b_r0 + 32 * j, p.N - 1);
          const float4 v = *reinterpret_cast<const float4*>(p.B + (long long)row * p.ldb + kc_);
          rb[sl][j] = kin ? v : zero4();
        }
      } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int row = n0 + b_r0 + 32 * j;
          const int valid = (row < p.N) ? min(4, max(0, p.K - k)) : 0;
          rb[sl][j] = valid ? ldg4(p.B + (long long)row * p.ldb + k, valid) : zero4();
        }
      }
    } else if constexpr (BMODE == 1) {
      const int ncol = n0 + 4 * b_mc;
      if constexpr (VEC) {
        const int nc_ = min(ncol, p.N - 4);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int k = kbase + (PREC ? 2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1) : b_k0 + B_KSTEP * j);
          const bool kin = KFULL || k < p.K;
          const float4 v = *reinterpret_cast<const float4*>(p.B + (long long)(kin ? k : 0) * p.ldb + nc_);
          rb[sl][j] = kin ? v : zero4();
        }
      } else {
        const int vn = min(4, max(0, p.N - ncol));
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int k = kbase + (PREC ? 2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1) : b_k0 + B_KSTEP * j);
          rb[sl][j] = (k < p.K && vn) ? ldg4(p.B + (long long)k * p.ldb + ncol, vn) : zero4();
        }
      }
    } else if constexpr (BMODE == 2) {  // weight (O, KH, KW, C) read as [k = tap*O + o][n = c]
      const int ncol = min(n0 + 4 * b_mc, p.N - 4);
      const int taps = p.cKH * p.cKW;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = kbase + (PREC ? 2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1) : b_k0 + B_KSTEP * j);
        const bool kin = KFULL || k < p.K;
        const int kq = kin ? k : 0;
        const int tap = kq / p.cO, o = kq - tap * p.cO;
        const float4 v = *reinterpret_cast<const float4*>(p.B + ((long long)o * taps + tap) * p.cC + ncol);
        rb[sl][j] = kin ? v : zero4();
      }
    } else {  // BMODE == 3: wgrad im2col; k = output position, n = tap*C + c
      const int ncol = min(n0 + 4 * b_mc, p.N - 4);
      const int tap = ncol / p.cC;
      const int c = ncol - tap * p.cC;
      const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = kbase + (PREC ? 2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1) : b_k0 + B_KSTEP * j);
        const bool kin = KFULL || k < p.K;
        const int kq = kin ? k : 0;
        const int ox = kq % p.cOW;
        const int tq = kq / p.cOW;
        const int oy = tq % p.cOH;
        const int n = tq / p.cOH;
        const int iy = oy * p.cStride - p.cPad + kh, ix = ox * p.cStride - p.cPad + kw;
        const bool ok = kin && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
        const int yy = min(max(iy, 0), p.cH - 1), xx = min(max(ix, 0), p.cW - 1);
        const float4 v = *reinterpret_cast<const float4*>(p.B + (((long long)n * p.cH + yy) * p.cW + xx) * p.cC + c);
        rb[sl][j] = ok ? v : zero4();
      }
    }
  };

  auto store_tiles = [&](int buf, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    if constexpr (PREC == 0) {
      float* Ad = As[buf];
      float* Bd = Bs[buf];
      if constexpr (A_KC) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
          *reinterpret_cast<float4*>(&Ad[(a_r0 + 32 * j) * A_LD + 4 * a_kc]) = ra[sl][j];
      } else {
#pragma unroll
        for (int j = 0; j < NA; ++j)
          *reinterpret_cast<float4*>(&Ad[(a_k0 + A_KSTEP * j) * A_LD + 4 * a_mc]) = ra[sl][j];
      }
      if constexpr (B_KC) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
          *reinterpret_cast<float4*>(&Bd[(b_r0 + 32 * j) * B_LD + 4 * b_kc]) = rb[sl][j];
      } else {
#pragma unroll
        for (int j = 0; j < NB; ++j)
          *reinterpret_cast<float4*>(&Bd[(b_k0 + B_KSTEP * j) * B_LD + 4 * b_mc]) = rb[sl][j];
      }
    } else {
      unsigned short* Ah = reinterpret_cast<unsigned short*>(As[buf]);
      unsigned short* Al = Ah + A_PL;
      unsigned short* Bh = reinterpret_cast<unsigned short*>(Bs[buf]);
      unsigned short* Bl = Bh + B_PL;
      if constexpr (A_KC) {          // 4 consecutive k of one row -> one 8-byte store per plane
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          unsigned h0, l0, h1, l1;
          split2(ra[sl][j].x, ra[sl][j].y, h0, l0);
          split2(ra[sl][j].z, ra[sl][j].w, h1, l1);
          const int o = (a_r0 + 32 * j) * PL_LD + 4 * a_kc;
          *reinterpret_cast<uint2*>(Ah + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Al + o) = make_uint2(l0, l1);
        }
      } else {                       // rows k, k+1 of 4 consecutive m: "pair-major" plane [k/2][m] of 32-bit (k, k+1)
                                     // words -> one conflict-free 16-byte store per plane (the [m][k] image would
                                     // put the 16 lanes of a store on 2 banks)
        unsigned* Ah32 = reinterpret_cast<unsigned*>(Ah);
#pragma unroll
        for (int jj = 0; jj < NA / 2; ++jj) {
          unsigned h[4], l[4];
          split2(ra[sl][2 * jj].x, ra[sl][2 * jj + 1].x, h[0], l[0]);
          split2(ra[sl][2 * jj].y, ra[sl][2 * jj + 1].y, h[1], l[1]);
          split2(ra[sl][2 * jj].z, ra[sl][2 * jj + 1].z, h[2], l[2]);
          split2(ra[sl][2 * jj].w, ra[sl][2 * jj + 1].w, h[3], l[3]);
          const int o = (a_k0 + A_KSTEP * jj) * PMA_LD + 4 * a_mc;
          *reinterpret_cast<uint4*>(Ah32 + o) = make_uint4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<uint4*>(Ah32 + A_PL / 2 + o) = make_uint4(l[0], l[1], l[2], l[3]);
        }
      }
      if constexpr (B_KC) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          unsigned h0, l0, h1, l1;
          split2(rb[sl][j].x, rb[sl][j].y, h0, l0);
          split2(rb[sl][j].z, rb[sl][j].w, h1, l1);
          const int o = (b_r0 + 32 * j) * PL_LD + 4 * b_kc;
          *reinterpret_cast<uint2*>(Bh + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Bl + o) = make_uint2(l0, l1);
        }
      } else {
        unsigned* Bh32 = reinterpret_cast<unsigned*>(Bh);
#pragma unroll
        for (int jj = 0; jj < NB / 2; ++jj) {
          unsigned h[4], l[4];
          split2(rb[sl][2 * jj].x, rb[sl][2 * jj + 1].x, h[0], l[0]);
          split2(rb[sl][2 * jj].y, rb[sl][2 * jj + 1].y, h[1], l[1]);
          split2(rb[sl][2 * jj].z, rb[sl][2 * jj + 1].z, h[2], l[2]);
          split2(rb[sl][2 * jj].w, rb[sl][2 * jj + 1].w, h[3], l[3]);
          const int o = (b_k0 + B_KSTEP * jj) * PMB_LD + 4 * b_mc;
          *reinterpret_cast<uint4*>(Bh32 + o) = make_uint4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<uint4*>(Bh32 + B_PL / 2 + o) = make_uint4(l[0], l[1], l[2], l[3]);
        }
      }
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;

  auto compute_groups = [&](int buf, int g0, int g1) {
    if constexpr (PREC == 1) {
      // groups 0,1 <-> k-step 0 ; groups 2,3 <-> k-step 1 (two 16-deep bf16 steps per 32-k tile)
      const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As[buf]);
      const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs[buf]);
      const int ks = g0 >> 1;
      bf16x8 ahi[TI], alo[TI], bhi[TJ], blo[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if constexpr (A_KC) {
          const int ao = (wm * WTM + 32 * i + l31) * PL_LD + ks * 16 + 8 * lh;
          ahi[i] = *reinterpret_cast<const bf16x8*>(Ah + ao);
          alo[i] = *reinterpret_cast<const bf16x8*>(Ah + A_PL + ao);
        } else {
          const unsigned* A32 = reinterpret_cast<const unsigned*>(Ah);
          const int ao = (ks * 8 + 4 * lh) * PMA_LD + wm * WTM + 32 * i + l31;
          const uint4 h4 = make_uint4(A32[ao], A32[ao + PMA_LD], A32[ao + 2 * PMA_LD], A32[ao + 3 * PMA_LD]);
          const uint4 l4 = make_uint4(A32[A_PL / 2 + ao], A32[A_PL / 2 + ao + PMA_LD], A32[A_PL / 2 + ao + 2 * PMA_LD],
                                      A32[A_PL / 2 + ao + 3 * PMA_LD]);
          ahi[i] = __builtin_bit_cast(bf16x8, h4);
          alo[i] = __builtin_bit_cast(bf16x8, l4);
        }
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        if constexpr (B_KC) {
          const int bo = (wn * WTN + 32 * j + l31) * PL_LD + ks * 16 + 8 * lh;
          bhi[j] = *reinterpret_cast<const bf16x8*>(Bh + bo);
          blo[j] = *reinterpret_cast<const bf16x8*>(Bh + B_PL + bo);
        } else {
          const unsigned* B32 = reinterpret_cast<const unsigned*>(Bh);
          const int bo = (ks * 8 + 4 * lh) * PMB_LD + wn * WTN + 32 * j + l31;
          const uint4 h4 = make_uint4(B32[bo], B32[bo + PMB_LD], B32[bo + 2 * PMB_LD], B32[bo + 3 * PMB_LD]);
          const uint4 l4 = make_uint4(B32[B_PL / 2 + bo], B32[B_PL / 2 + bo + PMB_LD], B32[B_PL / 2 + bo + 2 * PMB_LD],
                                      B32[B_PL / 2 + bo + 3 * PMB_LD]);
          bhi[j] = __builtin_bit_cast(bf16x8, h4);
          blo[j] = __builtin_bit_cast(bf16x8, l4);
        }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
        }
      return;
    }
    const float* Ar = As[buf];
    const float* Br = Bs[buf];
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      float af[TI][4], bf[TJ][4];
      const int kq = 8 * g + 4 * lh;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int row = wm * WTM + i * 32 + l31;
        if constexpr (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(&Ar[row * A_LD + kq]);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[i][s] = Ar[(kq + s) * A_LD + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = wn * WTN + j * 32 + l31;
        if constexpr (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(&Br[col * B_LD + kq]);
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = Br[(kq + s) * B_LD + col];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };

  // optional fused bias gradient (a_mode 1 = wgrad of nn.Linear, A = dY stored [tokens][N_out]):
  // colsum_out[m] += sum_k A[m][k] over this block's k range, from the A registers on their way to LDS.  Only the
  // tn == 0 blocks take part, so an address receives split_k atomic adds (<= 64); the same sums fused into the dgrad
  // (a_mode 0, one add per M-tile: 680 per address) serialised on the float atomics and cost 30 % of the step.
  // All NA loads of a thread sit in one 4-wide m chunk, so one float4 per thread carries the partial sums.
  const bool do_colsum = (AMODE == 1) && p.colsum_out != nullptr && tn == 0;
  float4 csum = zero4();
  auto colsum_tile = [&](auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    if constexpr (AMODE == 1) {
#pragma unroll
      for (int j = 0; j < NA; ++j) { csum.x += ra[sl][j].x; csum.y += ra[sl][j].y; csum.z += ra[sl][j].z; csum.w += ra[sl][j].w; }
    }
  };

  // software pipeline: tile r (relative to kt_begin) travels in register slot r % DEPTH; LDS is double-buffered.
  // step r: MFMAs of tile r from LDS[cur] | tile r+1: registers -> LDS[cur^1] | barrier | tile r+1+DEPTH: issue loads
  auto pipeline_step = [&](int t, int cur, auto NEXT) {            // NEXT = slot of tile t+1
    compute_groups(cur, 0, 2);
    if (t + 1 < kt_end) {
      if (do_colsum) colsum_tile(NEXT);
      store_tiles(cur ^ 1, NEXT);
    }
    compute_groups(cur, 2, 4);
    __syncthreads();                                               // all reads of `cur` and writes of `cur^1` are done
    if (t + 1 + DEPTH < kt_end) load_tiles(t + 1 + DEPTH, NEXT);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1 % DEPTH>;
  using S2 = std::integral_constant<int, 2 % DEPTH>;
  load_tiles(kt_begin, S0{});
  if constexpr (DEPTH >= 2)
    if (kt_begin + 1 < kt_end) load_tiles(kt_begin + 1, S1{});
  if constexpr (DEPTH == 3)
    if (kt_begin + 2 < kt_end) load_tiles(kt_begin + 2, S2{});
  if (do_colsum) colsum_tile(S0{});
  store_tiles(0, S0{});
  if (kt_begin + DEPTH < kt_end) load_tiles(kt_begin + DEPTH, S0{});
  __syncthreads();
  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; kt += DEPTH) {
    pipeline_step(kt, cur, S1{});
    cur ^= 1;
    if constexpr (DEPTH >= 2) {
      if (kt + 1 >= kt_end) break;
      pipeline_step(kt + 1, cur, S2{});
      cur ^= 1;
    }
    if constexpr (DEPTH == 3) {
      if (kt + 2 >= kt_end) break;
      pipeline_step(kt + 2, cur, S0{});
      cur ^= 1;
    }
  }

  if constexpr (AMODE == 1) {
    if (do_colsum) {                                               // block-uniform
      // the main loop ended on a barrier: As is free.  thread t holds chunk a_mc = t % A_CH; fold the 256 / A_CH rows
      float4* red = reinterpret_cast<float4*>(&As[0][0]);
      red[t] = csum;
      __syncthreads();
      if (t < A_CH) {
        float4 r = red[t];
#pragma unroll
        for (int i = 1; i < 256 / A_CH; ++i) { const float4 u = red[t + A_CH * i]; r.x += u.x; r.y += u.y; r.z += u.z; r.w += u.w; }
        const int m = m0 + 4 * t;
        // (a clamped tail chunk of the vector path re-reads valid columns: it must not be counted)
        if (m < p.M) atomicAdd(p.colsum_out + m, r.x);
        if (m + 1 < p.M) atomicAdd(p.colsum_out + m + 1, r.y);
        if (m + 2 < p.M) atomicAdd(p.colsum_out + m + 2, r.z);
        if (m + 3 < p.M) atomicAdd(p.colsum_out + m + 3, r.w);
      }
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  uint64_t seed = 0, step = 0;
  const bool drop = p.drop_thresh != 0;
  if (drop) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  const bool atomic = p.split_k > 1;
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  // Interior tiles with at most one extra epilogue operand take a branch-free form: the row base of accumulator register r is
  // wave-uniform (SGPR pointer arithmetic), the lane part is one 32-bit offset shared by the stores and by the operand
  // (host-checked: ldr == ldc, ldm == ldc), and the operand's 16 values are requested before the first store.  The generic
  // loop below (per-element conditions, 64-bit addresses) keeps the edge tiles, dropout and combined epilogues.
  const int n_extra = (p.residual != nullptr) + (p.mask_src != nullptr) + (p.accumulate != 0) + (drop ? 1 : 0);
  const bool same_ld = (!p.residual || p.ldr == p.ldc) && (!p.mask_src || p.ldm == p.ldc);
  if (interior && !drop && n_extra <= 1 && same_ld && (long long)BM * p.ldc < (1ll << 30)) {
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm_s = wave_s >> 1, wn_s = wave_s & 1;
    const int ldc = (int)p.ldc;
    const float floor_v = p.relu ? 0.f : -INFINITY;
    const float* xb = p.residual ? p.residual : p.mask_src ? p.mask_src : (p.accumulate && !atomic) ? p.C : nullptr;
    const int kind = atomic ? 1 : p.residual ? 2 : p.mask_src ? 3 : p.accumulate ? 4 : 0;     // block-uniform
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int colb = n0 + wn_s * WTN + j * 32;                                   // uniform
        const long long base = (long long)(m0 + wm_s * WTM + i * 32) * ldc + colb;    // uniform
        float* c0 = p.C + base;
        const unsigned lo = (unsigned)(4 * lh * ldc + l31);
        const float sc = p.scale ? p.scale[colb + l31] : 1.f;
        const float bi = p.bias ? p.bias[colb + l31] : 0.f;
        if (kind == 1) {
          const float b0 = split == 0 ? bi : 0.f;                                  // the bias rides with the first k-split
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(c0 + (lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * ldc)), acc[i][j][r] + b0);
        } else if (kind == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c0[lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * ldc)] = fmaxf(fmaf(acc[i][j][r], sc, bi), floor_v);
        } else {
          const float* x0 = xb + base;
          float xv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) xv[r] = x0[lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * ldc)];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = fmaf(acc[i][j][r], sc, bi);
            if (kind == 2) v = fmaxf(v + xv[r], floor_v);
            else if (kind == 3) v = xv[r] != 0.f ? fmaxf(v, floor_v) * p.mask_scale : 0.f;
            else v = fmaxf(v, floor_v) + xv[r];
            c0[lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * ldc)] = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      if (!interior && col >= p.N) continue;
      const float sc = p.scale ? p.scale[col] : 1.f;
      const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (!interior && row >= p.M) continue;
        float v = acc[i][j][r];
        float* cp = p.C + (long long)row * p.ldc + col;
        if (atomic) { atomicAdd(cp, split == 0 ? v + bi : v); continue; }     // the bias rides with the first k-split
        v = v * sc + bi;
        if (p.residual) v += p.residual[(long long)row * p.ldr + col];
        if (p.relu) v = fmaxf(v, 0.f);
        if (drop) v = cape_keep(seed, step, p.rng_stream, (uint64_t)row * (uint64_t)p.N + col, p.drop_thresh) ? v * p.inv_keep : 0.f;
        if (p.mask_src) v = p.mask_src[(long long)row * p.ldm + col] != 0.f ? v * p.mask_scale : 0.f;
        if (p.accumulate) v += *cp;
        *cp = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Skinny products (M <= 64 rows: the cached decode step multiplies 32 token rows by every weight of the decoder).  A
// 64x64 MFMA tile would run such a product on N/64 blocks of the 256 CUs with eight serial k-tiles of latency each
// (10 us measured for 32x256x256); here a block owns SK_COLS output columns, stages the activation rows and its weight
// rows in LDS (row stride K+4 floats: the 8 rows a wave touches fall on different banks) in chunks of <= 256 k, and every
// thread carries one (row, column) dot product in plain fp32 FMAs -- exact fp32, no reductions, N/8 blocks.
// ---------------------------------------------------------------------------------------------
constexpr int SK_COLS = 8, SK_KC = 256;

__global__ void __launch_bounds__(512) gemm_skinny_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float xs[64 * (SK_KC + 4)];
  __shared__ __attribute__((aligned(16))) float ws[SK_COLS * (SK_KC + 4)];
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * SK_COLS;
  const int col = t & (SK_COLS - 1), row = t >> 3;                // 512 threads = 64 rows x 8 columns
  const int LD = SK_KC + 4;
  float acc = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += SK_KC) {
    const int kc = min(SK_KC, p.K - k0);                          // multiple of 4 (host-checked)
    const int kq = kc >> 2;
    if (k0) __syncthreads();
    for (int i = t; i < p.M * kq; i += 512) {
      const int r = i / kq, c = (i - r * kq) * 4;
      *reinterpret_cast<float4*>(&xs[r * LD + c]) = *reinterpret_cast<const float4*>(p.A + (long long)r * p.lda + k0 + c);
    }
    for (int i = t; i < SK_COLS * kq; i += 512) {
      const int r = i / kq, c = (i - r * kq) * 4;
      const int n = min(n0 + r, p.N - 1);
      *reinterpret_cast<float4*>(&ws[r * LD + c]) = *reinterpret_cast<const float4*>(p.B + (long long)n * p.ldb + k0 + c);
    }
    __syncthreads();
    if (row < p.M) {
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
  const int n = n0 + col;
  if (row < p.M && n < p.N) {
    float v = acc * (p.scale ? p.scale[n] : 1.f) + (p.bias ? p.bias[n] : 0.f);
    if (p.residual) v += p.residual[(long long)row * p.ldr + n];
    if (p.relu) v = fmaxf(v, 0.f);
    float* cp = p.C + (long long)row * p.ldc + n;
    if (p.accumulate) v += *cp;
    *cp = v;
  }
}

template <int BM, int BN>
int launch_mode(const GemmP& p, int a_mode, int b_mode, bool vec, int prec, dim3 grid, hipStream_t s) {
  const bool kfull = vec && (p.K % BK == 0);
#define CASE(AM, BM_)                                                                      \
  if (a_mode == AM && b_mode == BM_) {                                                     \
    if (vec && prec == 1) {                                                                \
      if (kfull) hipLaunchKernelGGL((gemm_kernel<BM, BN, AM, BM_, true, 1, true>), grid, dim3(256), 0, s, p);   \
      else hipLaunchKernelGGL((gemm_kernel<BM, BN, AM, BM_, true, 1, false>), grid, dim3(256), 0, s, p);        \
      return 0;                                                                            \
    }                                                                                      \
    if (vec && kfull) hipLaunchKernelGGL((gemm_kernel<BM, BN, AM, BM_, true, 0, true>), grid, dim3(256), 0, s, p);   \
    else if (vec) hipLaunchKernelGGL((gemm_kernel<BM, BN, AM, BM_, true, 0, false>), grid, dim3(256), 0, s, p);      \
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, AM, BM_, false, 0, false>), grid, dim3(256), 0, s, p);              \
    return 0;                                                                              \
  }
  CASE(0, 0) CASE(2, 0) CASE(0, 1) CASE(3, 2) CASE(1, 1) CASE(1, 3)
#undef CASE
  return cape_set_error("cape_gemm_f32: unsupported (a_mode=%d, b_mode=%d)", a_mode, b_mode);
}

}  // namespace

extern "C" int cape_gemm_f32(const cape_gemm_desc* d, cape_stream_t stream) {
  CAPE_REQUIRE(d != nullptr, "cape_gemm_f32: null descriptor");
  CAPE_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "cape_gemm_f32: negative size");
  if (d->M == 0 || d->N == 0) return 0;
  CAPE_REQUIRE(d->A && d->B && d->C, "cape_gemm_f32: null operand");
  CAPE_REQUIRE(d->split_k >= 1, "cape_gemm_f32: split_k must be >= 1");
  if (d->split_k > 1)
    CAPE_REQUIRE(!d->scale && !d->residual && !d->relu && d->dropout_p == 0.f,
                 "cape_gemm_f32: split_k > 1 allows no epilogue op besides the bias and accumulation");
  if (d->a_mode == 2 || d->a_mode == 3 || d->b_mode == 2 || d->b_mode == 3) {
    CAPE_REQUIRE(d->cC % 4 == 0 && d->cO % 4 == 0, "cape_gemm_f32: conv channels must be multiples of 4 (C=%d, O=%d)", d->cC, d->cO);
    CAPE_REQUIRE(d->cStride >= 1 && d->cKH >= 1 && d->cKW >= 1, "cape_gemm_f32: bad conv geometry");
    const long long taps = (long long)d->cKH * d->cKW;
    if (d->a_mode == 2) CAPE_REQUIRE(d->K == taps * d->cC && d->M == (long long)d->cN * d->cOH * d->cOW, "cape_gemm_f32: conv-fwd shape mismatch");
    if (d->a_mode == 3) CAPE_REQUIRE(d->K == taps * d->cO && d->M == (long long)d->cN * d->cH * d->cW && d->N == d->cC, "cape_gemm_f32: conv-dgrad shape mismatch");
    if (d->b_mode == 3) CAPE_REQUIRE(d->N == taps * d->cC && d->K == (long long)d->cN * d->cOH * d->cOW && d->M == d->cO, "cape_gemm_f32: conv-wgrad shape mismatch");
    CAPE_REQUIRE((reinterpret_cast<uintptr_t>(d->A) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->B) & 15) == 0,
                 "cape_gemm_f32: conv operands must be 16-byte aligned");
  }
  if (d->dropout_p > 0.f) CAPE_REQUIRE(d->rng_state != nullptr && d->dropout_p < 1.f, "cape_gemm_f32: dropout needs rng_state and p < 1");

  GemmP p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.B = d->B; p.ldb = d->ldb; p.C = d->C; p.ldc = d->ldc;
  p.cN = d->cN; p.cH = d->cH; p.cW = d->cW; p.cC = d->cC; p.cKH = d->cKH; p.cKW = d->cKW;
  p.cStride = d->cStride; p.cPad = d->cPad; p.cOH = d->cOH; p.cOW = d->cOW; p.cO = d->cO;
  p.scale = d->scale; p.bias = d->bias; p.residual = d->residual; p.ldr = d->ldr;
  p.relu = d->relu; p.accumulate = d->accumulate; p.split_k = d->split_k;
  p.colsum_out = d->colsum_out;
  if (d->colsum_out) CAPE_REQUIRE(d->a_mode == 1, "cape_gemm_f32: colsum_out needs a_mode 1 (the wgrad product)");
  p.drop_thresh = d->dropout_p > 0.f ? cape_drop_threshold(d->dropout_p) : 0u;
  p.inv_keep = d->dropout_p > 0.f ? 1.f / (1.f - d->dropout_p) : 1.f;
  p.rng_state = d->rng_state; p.rng_stream = d->rng_stream;

  p.Bpack = d->B_packed;
  p.mask_src = d->mask_src; p.ldm = d->ldm; p.mask_scale = d->mask_scale;
  p.bdiv = d->batch_div > 0 ? d->batch_div : 1;
  p.sA0 = d->sA0; p.sA1 = d->sA1; p.sB0 = d->sB0; p.sB1 = d->sB1; p.sC0 = d->sC0; p.sC1 = d->sC1;
  if (d->batch > 1)
    CAPE_REQUIRE(d->batch <= 65535 && d->split_k == 1 && !d->bias && !d->scale && !d->residual && !d->mask_src && !d->colsum_out &&
                     d->dropout_p == 0.f && (d->a_mode == 0 || d->a_mode == 1) && (d->b_mode == 0 || d->b_mode == 1),
                 "cape_gemm_f32: batched launches take dense modes, no epilogue vectors, split_k 1, batch <= 65535");
  if (d->mask_src) CAPE_REQUIRE(d->split_k == 1, "cape_gemm_f32: mask_src needs split_k == 1");
  CAPE_REQUIRE(d->precision == 0 || d->precision == 1, "cape_gemm_f32: precision must be 0 (fp32) or 1 (bf16x3)");
  // register-stationary weights (gemm_rs.hip): dense A against a <= 256-deep weight, the token products of the transformer
  if (d->precision == 1 && d->batch <= 1 && cape_gemm_rs_eligible(p, d->a_mode, d->b_mode)) return cape_gemm_rs_launch(p, d->b_mode, as_stream(stream));

  // vector path: every 16-byte load must be aligned and stay inside its row
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool vec = al16(d->A) && al16(d->B);
  if (d->a_mode == 0) vec = vec && (d->lda % 4 == 0) && (d->K % 4 == 0);
  if (d->a_mode == 1) vec = vec && (d->lda % 4 == 0) && (d->M % 4 == 0) && d->M >= 4;
  if (d->b_mode == 0) vec = vec && (d->ldb % 4 == 0) && (d->K % 4 == 0);
  if (d->b_mode == 1) vec = vec && (d->ldb % 4 == 0) && (d->N % 4 == 0) && d->N >= 4;
  if (d->b_mode == 2 || d->b_mode == 3) vec = vec && (d->N % 4 == 0) && d->N >= 4;
  if (d->batch > 1) vec = vec && ((d->sA0 | d->sA1 | d->sB0 | d->sB1) % 4 == 0);
  if ((d->a_mode >= 2 || d->b_mode >= 2) && !vec) return cape_set_error("cape_gemm_f32: conv modes need the aligned vector path");
  CAPE_REQUIRE(d->precision == 0 || d->precision == 1, "cape_gemm_f32: precision must be 0 (fp32) or 1 (bf16x3)");
  // skinny products: M <= 64 rows of a dense NT product go to the FMA kernel (exact fp32 in either precision mode)
  if (d->a_mode == 0 && d->b_mode == 0 && d->M <= 64 && d->batch <= 1 && vec && d->split_k == 1 && d->dropout_p == 0.f && !d->colsum_out &&
      !d->mask_src && (d->N + SK_COLS - 1) / SK_COLS < (1 << 30)) {
    static const bool off = getenv("CAPE_GEMM_NO_SKINNY") != nullptr;      // tuning switch
    if (!off) {
      hipLaunchKernelGGL(gemm_skinny_kernel, dim3((d->N + SK_COLS - 1) / SK_COLS), dim3(512), 0, as_stream(stream), p);
      CAPE_LAUNCH_CHECK("cape_gemm_f32(skinny)");
      return 0;
    }
  }

  // tile choice.  Measured on MI355X (tools/gemm_bench.py): with the 64-cycle fp32 MFMA step the 64x64 tile (4 blocks
  // per CU, 4 waves/SIMD) is never slower than 128x128 and much better balanced on this model's shapes
  // (M = 43520 = 340 x 128 gives 680 big tiles on 512 slots = 1.33 rounds; 2720 small tiles on 1024 slots waste far
  // less): 43520x256x256 57 -> 77 TF/s, x1024 74 -> 93 TF/s, 4096^3 118 = 118 TF/s.  128x128 stays available for tuning.
  // ... except for deep contractions: with K >= 1024 the 128x128 tile's halved L2 -> L1 traffic per flop wins once there
  // are enough tiles to cover the chip (measured: 43520x256x1024 134 -> 111 us, 1024x256x43520 split 16 118 -> 111 us;
  // but 256x256x43520 split 64, one 128-tile block per CU: 36 -> 40 us, hence the floor on the output size; and
  // 1024x256x6400 split 16 (400 deep per split): 27.7 -> 28.9 us, hence the depth is counted per k-split)
  const long long t128 = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128) * d->split_k;
  bool big = d->K / d->split_k >= 1024 && t128 >= 256 && (long long)d->M * d->N >= 256 * 1024 && d->precision == 1 && vec;
  {
    static const char* force = getenv("CAPE_GEMM_TILE");      // tuning override: 64 or 128
    if (force && force[0] == '1') big = true;
    if (force && force[0] == '6') big = false;
  }
  const int BMv = big ? 128 : 64;
  p.tilesM = (d->M + BMv - 1) / BMv;
  p.tilesN = (d->N + BMv - 1) / BMv;
  const long long ntiles = (long long)p.tilesM * p.tilesN;
  CAPE_REQUIRE(ntiles < (1ll << 31), "cape_gemm_f32: too many tiles");
  CAPE_REQUIRE(ntiles * d->split_k < (1ll << 31), "cape_gemm_f32: grid too large");
  dim3 grid((unsigned)(ntiles * d->split_k), (unsigned)(d->batch > 1 ? d->batch : 1));
  int rc = big ? launch_mode<128, 128>(p, d->a_mode, d->b_mode, vec, d->precision, grid, as_stream(stream))
               : launch_mode<64, 64>(p, d->a_mode, d->b_mode, vec, d->precision, grid, as_stream(stream));
  if (rc) return rc;
  CAPE_LAUNCH_CHECK("cape_gemm_f32");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// column sums: out[n] (+)= sum_m X[m][n].  grid (ceil(N/64), row splits); one column per lane,
// 4 waves of a block stride over rows; partials combined through LDS then one atomic per column.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) colsum_kernel(const float* X, long long ldx, long long batch_stride, int rows_per_batch,
                                                      long long M, int N, float* out, long long rows_per_block) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (long long r = r0 + w; r < r1; r += 4) {
      const long long b = r / rows_per_batch, rr = r - b * rows_per_batch;
      s += X[b * batch_stride + rr * ldx + col];
    }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < N) atomicAdd(&out[col], part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

// 16-byte form (N % 4 == 0, aligned rows): a wave covers 256 columns of a row (1 KB contiguous); a block is 16 waves
// (16 rows per step, 4 rows in flight per wave).  Few, fat blocks on purpose: the final float atomics of all blocks hit
// the same N addresses and serialise per address (~50 ns each, measured: 340 blocks -> 17 us of pure contention), so the
// row range is cut into at most ~96 blocks and each block adds once per column.
__global__ void __launch_bounds__(1024) colsum_vec_kernel(const float* X, long long ldx, long long batch_stride,
                                                           int rows_per_batch, long long M, int N, float* out,
                                                           long long rows_per_block) {
  __shared__ float4 part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 4;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto rowptr = [&](long long r) {
    const long long b = r / rows_per_batch, rr = r - b * rows_per_batch;
    return reinterpret_cast<const float4*>(X + b * batch_stride + rr * ldx + col);
  };
  if (col < N) {
    long long r = r0 + w;
    for (; r + 48 < r1; r += 64) {
      const float4 a = *rowptr(r), b = *rowptr(r + 16), c = *rowptr(r + 32), d = *rowptr(r + 48);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
      s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; r < r1; r += 16) {
      const float4 a = *rowptr(r);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  part[w][lane] = make_float4(s0.x + s1.x + s2.x + s3.x, s0.y + s1.y + s2.y + s3.y, s0.z + s1.z + s2.z + s3.z,
                              s0.w + s1.w + s2.w + s3.w);
  __syncthreads();
  if (w == 0 && col < N) {
    float4 t = part[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = part[k][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    atomicAdd(&out[col + 0], t.x);
    atomicAdd(&out[col + 1], t.y);
    atomicAdd(&out[col + 2], t.z);
    atomicAdd(&out[col + 3], t.w);
  }
}

extern "C" int cape_colsum_f32(const float* X, long long ldx, int nbatch, long long batch_stride, int M, int N, float* out,
                               int accumulate, cape_stream_t stream) {
  CAPE_REQUIRE(X && out && M >= 0 && N > 0 && nbatch >= 1, "cape_colsum_f32: bad arguments");
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, as_stream(stream));
    if (e != hipSuccess) return cape_set_error("cape_colsum_f32: memset: %s", hipGetErrorString(e));
  }
  if (M == 0) return 0;
  const long long rows = (long long)M * nbatch;
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0) && (batch_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const int gx = vec ? (N / 4 + 63) / 64 : (N + 63) / 64;
  long long splits = vec ? (rows + 255) / 256 : (rows + 127) / 128;
  const long long max_splits = vec ? (96 + gx - 1) / gx : (1024 + gx - 1) / gx;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const long long rpb = (rows + splits - 1) / splits;
  if (vec)
    hipLaunchKernelGGL(colsum_vec_kernel, dim3(gx, (unsigned)splits), dim3(1024), 0, as_stream(stream), X, ldx, batch_stride, M,
                       rows, N, out, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, (unsigned)splits), dim3(256), 0, as_stream(stream), X, ldx, batch_stride, M, rows,
                       N, out, rpb);
  CAPE_LAUNCH_CHECK("cape_colsum_f32");
  return 0;
}
