// gemm_tile.h -- the tile body of the GEMM family (see gemm.hip for the design notes): shared by the single-product kernel
// (gemm.hip) and the grouped launch (gemm_group.hip).
#pragma once
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
// One output tile (or one k-split of it) of a product of the family: `p` = the product (uniform), `bid` = the block's index among
// the tilesM * tilesN * split_k blocks of that product (block ids beyond them return: grouped launches pad each product's range
// to a multiple of 8 so that `bid & 7` keeps naming the blocks that share an XCD).
template <int BM, int BN, int AMODE, int BMODE, bool VEC, int PREC, bool KFULL>
__device__ __forceinline__ void gemm_tile_body(const GemmP& p, const int bid) {
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
  // PREC 1: per operand and buffer two bf16 planes (hi, lo).  k-contiguous operands: [row][k] with an 80-byte row stride
  // (32 k x 2 B + 16 B pad: conflict-free ds_read_b128 of 8 consecutive k per lane); mn-contiguous operands: the swizzled
  // [k][mn] image of gemm_common.h tr_off (32 rows of BM / BN bf16), read with ds_read_b64_tr_b16 -- it fits in the same plane
  constexpr int PL_LD = 40;                                    // bf16 elements per row ([row][k] image)
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
  if (bid >= ntiles * p.split_k) return;
  {
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
        else { a_y[j] = y + p.cPad; a_x[j] = x + p.cPadX; }
      } else {
        a_n[j] = -1; a_y[j] = 0; a_x[j] = 0;
      }
    }
  }

  // conv gather fast paths (round 3).  The general forms below recompute (tap, channel) of every k index with integer divisions per
  // thread and k-tile and rebuild a 64-bit pixel address per row -- 35-75 % more time than the dense product of the same shape
  // (tools/conv_bench.py), in a loop that is bound by instruction issue.  When the channel count is a multiple of the 32-deep
  // k-tile, a k-tile lies inside ONE filter tap: (kh, kw, channel offset) is a wave-uniform state advanced once per k-tile, a row
  // keeps the element offset of its centre pixel, and a load address is that offset plus a uniform tap offset (one 64-bit add,
  // two compares).  Same loads, same arithmetic, bitwise identical results.
  constexpr bool A_CONV = (AMODE == 2 || AMODE == 3);
  const bool a_fast = A_CONV && KFULL && (AMODE == 2 ? (p.cC % BK == 0) : (p.cO % BK == 0 && p.cStride == 1));
  long long a_base[NA];
  int cv_kh = 0, cv_kw = 0, cv_c0 = 0;                         // tap and channel offset of the k-tile the NEXT load_tiles call reads
  if constexpr (A_CONV) {
    if (a_fast) {
      const int CC = (AMODE == 2) ? p.cC : p.cO;
      const int k0 = kt_begin * BK;
      const int tap = k0 / CC;
      cv_c0 = k0 - tap * CC;
      cv_kh = tap / p.cKW;
      cv_kw = tap - cv_kh * p.cKW;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const long long nn = max(a_n[j], 0);
        a_base[j] = (AMODE == 2) ? ((nn * p.cH + a_y[j]) * p.cW + a_x[j]) * p.cC + 4 * a_kc
                                 : ((nn * p.cOH + a_y[j]) * p.cOW + a_x[j]) * p.cO + 4 * a_kc;
      }
    }
  }
  // BMODE 2 (dgrad weight [k = tap*O + o][n = c], always with AMODE 3): per-thread constant part of the address
  int b_koff[NB];
  if constexpr (BMODE == 2) {
    const int ncol = min(n0 + 4 * b_mc, p.N - 4);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int kofs = PREC ? 2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1) : b_k0 + B_KSTEP * j;
      b_koff[j] = kofs * (p.cKHp * p.cKWp) * p.cC + ncol;       // (physical taps per output channel)
    }
  }
  // BMODE 3 (wgrad im2col, k = output position): power-of-two output extents turn the two divisions per load into shifts
  int b3_lw = -1, b3_lh = 0;
  if constexpr (BMODE == 3) {
    if ((p.cOW & (p.cOW - 1)) == 0 && (p.cOH & (p.cOH - 1)) == 0) { b3_lw = __ffs(p.cOW) - 1; b3_lh = __ffs(p.cOH) - 1; }
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
      if (a_fast) {
        const long long tap_off = ((long long)cv_kh * p.cW + cv_kw) * p.cC + cv_c0;          // uniform
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int iy = a_y[j] + cv_kh, ix = a_x[j] + cv_kw;
          const bool ok = (a_n[j] >= 0) && (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
          const float4 v = *reinterpret_cast<const float4*>(p.A + (ok ? a_base[j] + tap_off : (long long)(4 * a_kc)));
          ra[sl][j] = ok ? v : zero4();
        }
      } else {
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
      }
    } else if (a_fast) {  // AMODE == 3, stride 1, O % 32 == 0
      const long long tap_off = cv_c0 - ((long long)cv_kh * p.cOW + cv_kw) * p.cO;           // uniform
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int ty = a_y[j] - cv_kh, tx = a_x[j] - cv_kw;
        const bool ok = (a_n[j] >= 0) && (unsigned)ty < (unsigned)p.cOH && (unsigned)tx < (unsigned)p.cOW;
        const float4 v = *reinterpret_cast<const float4*>(p.A + (ok ? a_base[j] + tap_off : (long long)(4 * a_kc)));
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
    } else if (BMODE == 2 && a_fast) {  // the k-tile lies in one tap: (o0 * taps + tap) * C is uniform, the rest per-thread constant
      const long long tb = ((long long)cv_c0 * (p.cKHp * p.cKWp) + (p.cTapH0 + cv_kh * p.cTapHS) * p.cKWp + p.cTapW0 + cv_kw * p.cTapWS) * p.cC;
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[sl][j] = *reinterpret_cast<const float4*>(p.B + tb + b_koff[j]);
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
        int ox, oy, n;
        if (b3_lw >= 0) {
          ox = kq & (p.cOW - 1);
          oy = (kq >> b3_lw) & (p.cOH - 1);
          n = kq >> (b3_lw + b3_lh);
        } else {
          ox = kq % p.cOW;
          const int tq = kq / p.cOW;
          oy = tq % p.cOH;
          n = tq / p.cOH;
        }
        const int iy = oy * p.cStride - p.cPad + kh, ix = ox * p.cStride - p.cPad + kw;
        const bool ok = kin && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
        const int yy = min(max(iy, 0), p.cH - 1), xx = min(max(ix, 0), p.cW - 1);
        const float4 v = *reinterpret_cast<const float4*>(p.B + (((long long)n * p.cH + yy) * p.cW + xx) * p.cC + c);
        rb[sl][j] = ok ? v : zero4();
      }
    }
    if constexpr (A_CONV) {                                       // the next call reads the next k-tile (calls come in k order)
      if (a_fast) {
        cv_c0 += BK;
        if (cv_c0 == ((AMODE == 2) ? p.cC : p.cO)) {
          cv_c0 = 0;
          if (++cv_kw == p.cKW) { cv_kw = 0; ++cv_kh; }
        }
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
      } else {                       // 4 consecutive m of one k row: the plane is the plain [k][m] image (rows of BM bf16, 16-byte
                                     // chunks XOR-swizzled by the row, see tr_off) that ds_read_b64_tr_b16 turns into fragments
        char* Ahb = reinterpret_cast<char*>(Ah);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          unsigned h0, l0, h1, l1;
          split2(ra[sl][j].x, ra[sl][j].y, h0, l0);
          split2(ra[sl][j].z, ra[sl][j].w, h1, l1);
          const int o = tr_off<BM>(2 * (a_k0 + A_KSTEP * (j >> 1)) + (j & 1), a_mc >> 1) + 8 * (a_mc & 1);
          *reinterpret_cast<uint2*>(Ahb + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Ahb + 2 * A_PL + o) = make_uint2(l0, l1);
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
        char* Bhb = reinterpret_cast<char*>(Bh);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          unsigned h0, l0, h1, l1;
          split2(rb[sl][j].x, rb[sl][j].y, h0, l0);
          split2(rb[sl][j].z, rb[sl][j].w, h1, l1);
          const int o = tr_off<BN>(2 * (b_k0 + B_KSTEP * (j >> 1)) + (j & 1), b_mc >> 1) + 8 * (b_mc & 1);
          *reinterpret_cast<uint2*>(Bhb + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Bhb + 2 * B_PL + o) = make_uint2(l0, l1);
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
  // transposed fragment reads ([k][mn] images of the mn-contiguous operands): lane = 32 h + 16 c + 4 q + p supplies the address of
  // row q, columns 4 p .. 4 p + 3 of its 16-lane group's 4 x 16 block (column half c of the 32-wide fragment, k half h)
  const int tr_h = lane >> 5, tr_c = (lane >> 4) & 1, tr_q = (lane >> 2) & 3, tr_p = lane & 3;

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
          // [k][m] image: two transposed 4-row reads per plane deliver the lane's 8 consecutive k of its row (lane & 31)
          const char* Ab = reinterpret_cast<const char*>(Ah) + tr_off<BM>(ks * 16 + 8 * tr_h + tr_q, ((wm * WTM + 32 * i) >> 3) + 2 * tr_c + (tr_p >> 1)) + 8 * (tr_p & 1);
          ahi[i] = tr_read8(Ab, 4 * BM * 2);
          alo[i] = tr_read8(Ab + 2 * A_PL, 4 * BM * 2);
        }
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        if constexpr (B_KC) {
          const int bo = (wn * WTN + 32 * j + l31) * PL_LD + ks * 16 + 8 * lh;
          bhi[j] = *reinterpret_cast<const bf16x8*>(Bh + bo);
          blo[j] = *reinterpret_cast<const bf16x8*>(Bh + B_PL + bo);
        } else {
          const char* Bb = reinterpret_cast<const char*>(Bh) + tr_off<BN>(ks * 16 + 8 * tr_h + tr_q, ((wn * WTN + 32 * j) >> 3) + 2 * tr_c + (tr_p >> 1)) + 8 * (tr_p & 1);
          bhi[j] = tr_read8(Bb, 4 * BN * 2);
          blo[j] = tr_read8(Bb + 2 * B_PL, 4 * BN * 2);
        }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if (!p.single) {                                             // uniform; precision 2 keeps hi x hi only
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
          }
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
        const int kind_t = (kind == 2 && p.res_cols && colb >= p.res_cols) ? 0 : kind;      // uniform: this 32-column group takes no residual
        if (kind == 1) {
          const float b0 = split == 0 ? bi : 0.f;                                  // the bias rides with the first k-split
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(c0 + (lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * ldc)), acc[i][j][r] + b0);
        } else if (kind_t == 0) {
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
        if (p.residual && (!p.res_cols || col < p.res_cols)) v += p.residual[(long long)row * p.ldr + col];
        if (p.relu) v = fmaxf(v, 0.f);
        if (drop) v = cape_keep(seed, step, p.rng_stream, (uint64_t)row * (uint64_t)p.N + col, p.drop_thresh) ? v * p.inv_keep : 0.f;
        if (p.mask_src) v = p.mask_src[(long long)row * p.ldm + col] != 0.f ? v * p.mask_scale : 0.f;
        if (p.accumulate) v += *cp;
        *cp = v;
      }
    }
  }
}

}  // namespace
