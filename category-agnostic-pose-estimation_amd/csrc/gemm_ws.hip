// gemm_ws.hip -- weight-stationary bf16x3 GEMM for gfx950: C[M,N] = epi(A[M,K] * W[N,K]^T) where W is a weight
// whose bf16 (hi, lo) planes were split once per optimizer step (cape_split_planes) instead of once per tile.
//
// Why a second kernel.  PMC counters on the general kernel (gemm.hip) showed it bound by what surrounds the MFMAs,
// not by them: per 64x64x32 tile-step it converts 4096 fp32 values (VALU), writes 16 KB to LDS at ~85 B/clk and reads
// 32 KB back, for 192 MFMA cycles per SIMD -- LDS stores alone cost more than the matrix work.  Here
//   * the activation operand A never touches LDS: a wave owns 32 rows and loads its MFMA fragments straight from
//     global memory (lane = (row, k-half): 8 consecutive k = two 16-byte loads), splits them in registers -- every A
//     element is converted exactly once per N-tile instead of once per wave that needs it;
//   * the weight operand arrives pre-split: its bf16 planes go global -> registers -> LDS as plain 16-byte copies (no
//     VALU), 8 KB per k-tile for the whole block, and are read with conflict-free ds_read_b128;
//   * block = NW waves stacked along M (NW*32 x 64 tile): LDS traffic per MFMA is 1/3 of the general kernel's, there
//     is no LDS store of A, and the only barrier per k-tile guards the 8 KB weight tile.
// Measured (round 1, tools/gemm_bench.py): correct, but NOT faster than the general kernel on this model's shapes
// (43520x256x256: 46 us vs 39 us; 4096^3: 222 TF/s both).  Removing the A loads from the loop (timing experiment)
// took 4096^3 from 618 to 360 us, removing the weight path as well to 284 us: the row-per-lane fragment loads re-read
// A from L2 once per 64-column tile, and L2 -> L1 traffic (~10-14 TB/s effective on both kernels) is what bounds the
// family, not VALU, LDS or MFMA issue.  The lever is tile area (reuse per L2 byte), see DESIGN.md.  The training path
// therefore does not pass planes; the kernel stays as a tested option of the C ABI (cape_gemm_desc.B_hi/B_lo).
// a_mode 0 (dense rows), 2 (conv-forward gather) and 3 (conv-dgrad gather) are supported; the k index of a gather
// is kept wave-uniform per 16-deep step by requiring the channel count to be a multiple of 16.
#include <stdlib.h>
#include <type_traits>
#include "gemm_common.h"

namespace {

constexpr int WS_BN = 64;
constexpr int WS_PL = 40;                                          // bf16 per LDS row: 32 k + 8 pad (80-byte stride)

template <int NW, int AMODE>
__global__ void __launch_bounds__(NW * 64) gemm_ws_kernel(const GemmP p) {
  constexpr int BM = NW * 32, BN = WS_BN, NT = NW * 64;
  constexpr int CH = 256 / NT;                                     // 16-byte weight chunks per thread, plane and k-tile
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][BN * WS_PL];   // [buffer][plane][col][k]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware tile order (as gemm.hip): an XCD walks a contiguous run of tile ids, tiles of one A row panel adjacent
  int tile;
  {
    const int ntiles = p.tilesM * p.tilesN;
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ktiles = p.K / BK;                                     // host-checked: K % 32 == 0

  // ---- A addressing: this lane feeds row (m0 + 32*wave + l31), k offsets 8*lh .. 8*lh+7 of each 16-deep step
  const int row = m0 + wave * 32 + l31;
  const bool row_ok = row < p.M;
  const int rowc = row_ok ? row : p.M - 1;
  const float* arow = nullptr;
  int g_n = 0, g_y = 0, g_x = 0;
  if constexpr (AMODE == 0) {
    arow = p.A + (long long)rowc * p.lda + 8 * lh;
  } else {
    const int RW = (AMODE == 2) ? p.cOW : p.cW;
    const int RH = (AMODE == 2) ? p.cOH : p.cH;
    const int x = rowc % RW;
    const int tq = rowc / RW;
    const int y = tq % RH;
    g_n = tq / RH;
    if (AMODE == 2) { g_y = y * p.cStride - p.cPad; g_x = x * p.cStride - p.cPad; }
    else { g_y = y + p.cPad; g_x = x + p.cPad; }
  }

  // one 16-deep step of A: 8 consecutive k for this lane (two 16-byte loads)
  auto load_a = [&](int kt, int s, float4& x0, float4& x1) {
    const int k0 = kt * BK + 16 * s;                               // wave-uniform
    if constexpr (AMODE == 0) {
      const float4* q = reinterpret_cast<const float4*>(arow + k0);
      x0 = q[0]; x1 = q[1];
    } else if constexpr (AMODE == 2) {                             // k = tap*C + c, C % 16 == 0: one tap per step
      const int tap = k0 / p.cC, c0 = k0 - tap * p.cC;
      const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
      const int iy = g_y + kh, ix = g_x + kw;
      const bool ok = row_ok && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
      const int yy = min(max(iy, 0), p.cH - 1), xx = min(max(ix, 0), p.cW - 1);
      const float4* q = reinterpret_cast<const float4*>(p.A + (((long long)g_n * p.cH + yy) * p.cW + xx) * p.cC + c0 + 8 * lh);
      const float4 a = q[0], b = q[1];
      x0 = ok ? a : zero4(); x1 = ok ? b : zero4();
    } else {                                                       // dgrad: k = tap*O + o, O % 16 == 0
      const int tap = k0 / p.cO, o0 = k0 - tap * p.cO;
      const int kh = tap / p.cKW, kw = tap - kh * p.cKW;
      const int ty = g_y - kh, tx = g_x - kw;
      bool ok = row_ok && ty >= 0 && tx >= 0;
      int oy = ty, ox = tx;
      if (p.cStride != 1) {
        oy = ty / p.cStride; ox = tx / p.cStride;
        ok = ok && (oy * p.cStride == ty) && (ox * p.cStride == tx);
      }
      ok = ok && oy < p.cOH && ox < p.cOW;
      const int yy = min(max(oy, 0), p.cOH - 1), xx = min(max(ox, 0), p.cOW - 1);
      const float4* q = reinterpret_cast<const float4*>(p.A + (((long long)g_n * p.cOH + yy) * p.cOW + xx) * p.cO + o0 + 8 * lh);
      const float4 a = q[0], b = q[1];
      x0 = ok ? a : zero4(); x1 = ok ? b : zero4();
    }
  };

  // ---- weight tile: per plane 64 columns x 4 chunks of 8 bf16; chunk id -> (column, chunk)
  const unsigned short* bsrc_hi[CH];
  const unsigned short* bsrc_lo[CH];
  int bdst[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int q = t + NT * i;
    const int col = q >> 2, kc = q & 3;
    const long long o = (long long)min(n0 + col, p.N - 1) * p.ldp + 8 * kc;
    bsrc_hi[i] = p.Bhi + o;
    bsrc_lo[i] = p.Blo + o;
    bdst[i] = col * WS_PL + 8 * kc;
  }
  auto load_b = [&](int kt, uint4 (&bh)[CH], uint4 (&bl)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      bh[i] = *reinterpret_cast<const uint4*>(bsrc_hi[i] + kt * BK);
      bl[i] = *reinterpret_cast<const uint4*>(bsrc_lo[i] + kt * BK);
    }
  };
  auto store_b = [&](int buf, const uint4 (&bh)[CH], const uint4 (&bl)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      *reinterpret_cast<uint4*>(&Bs[buf][0][bdst[i]]) = bh[i];
      *reinterpret_cast<uint4*>(&Bs[buf][1][bdst[i]]) = bl[i];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- pipeline.  Per k-tile: split the A registers into bf16 fragments (the fp32 registers are dead afterwards), issue
  // the global loads of tile kt+1 into those same registers, run the MFMAs of tile kt while they fly, park the weight
  // chunks of tile kt+1 in the other LDS buffer, barrier.  A plain single-step loop on purpose: an unrolled
  // ping-pong of register sets made the compiler shuttle all 32 accumulators AGPR -> VGPR -> AGPR on every trip.
  float4 a[2][2];
  uint4 bh[CH], bl[CH];
  load_b(0, bh, bl);
  load_a(0, 0, a[0][0], a[0][1]);
  load_a(0, 1, a[1][0], a[1][1]);
  store_b(0, bh, bl);
  __syncthreads();
  int buf = 0;
  // MORE is a compile-time flag (the last k-tile is peeled) so that the A registers are provably dead after the split
  auto tile_step = [&](int kt, auto MORE) {
    constexpr bool more = decltype(MORE)::value;
    bf16x8 ahi[2], alo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned h[4], l[4];
      split2(a[s][0].x, a[s][0].y, h[0], l[0]);
      split2(a[s][0].z, a[s][0].w, h[1], l[1]);
      split2(a[s][1].x, a[s][1].y, h[2], l[2]);
      split2(a[s][1].z, a[s][1].w, h[3], l[3]);
      ahi[s] = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
      alo[s] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    }
    if constexpr (more) {
      load_b(kt + 1, bh, bl);
      load_a(kt + 1, 0, a[0][0], a[0][1]);
      load_a(kt + 1, 1, a[1][0], a[1][1]);
    }
    const unsigned short* Bh = &Bs[buf][0][0];
    const unsigned short* Bl = &Bs[buf][1][0];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int bo = (32 * j + l31) * WS_PL + 16 * s + 8 * lh;
        const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(Bh + bo);
        const bf16x8 blo = *reinterpret_cast<const bf16x8*>(Bl + bo);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[s], bhi, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], blo, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], bhi, acc[j], 0, 0, 0);
      }
    }
    if constexpr (more) {
      store_b(buf ^ 1, bh, bl);
      __syncthreads();
      buf ^= 1;
    }
  };
  for (int kt = 0; kt + 1 < ktiles; ++kt) tile_step(kt, std::true_type{});
  tile_step(ktiles - 1, std::false_type{});

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  uint64_t seed = 0, stp = 0;
  const bool drop = p.drop_thresh != 0;
  if (drop) { seed = p.rng_state[0]; stp = p.rng_state[1]; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 32 * j + l31;
    if (col >= p.N) continue;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (orow >= p.M) continue;
      float v = acc[j][r] * sc + bi;
      float* cp = p.C + (long long)orow * p.ldc + col;
      if (p.residual) v += p.residual[(long long)orow * p.ldr + col];
      if (p.relu) v = fmaxf(v, 0.f);
      if (drop) v = cape_keep(seed, stp, p.rng_stream, (uint64_t)orow * (uint64_t)p.N + col, p.drop_thresh) ? v * p.inv_keep : 0.f;
      if (p.accumulate) v += *cp;
      *cp = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight split: W (O, T, C) fp32 -> bf16 planes  P[o][t*C + c]  (as stored)  and/or  Pt[c][t*O + o]  (the operand
// of the transposed product: nn.Linear dgrad with T = 1, convolution dgrad with T = KH*KW).
// One block per 32 x 32 (o, c) tile of one tap; the transposed planes go through LDS so both outputs are coalesced.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split1(float x, unsigned short& hi, unsigned short& lo) {
  const __bf16 h = (__bf16)x;
  const __bf16 l = (__bf16)(x - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ W, int O, int T, int C,
                                                            unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                            unsigned short* __restrict__ hiT, unsigned short* __restrict__ loT) {
  __shared__ unsigned short th[32][33], tl[32][33];
  const int c0 = blockIdx.x * 32, o0 = blockIdx.y * 32, tap = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 8 rows per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o0 + ty + 8 * i, c = c0 + tx;
    unsigned short h = 0, l = 0;
    if (o < O && c < C) {
      const long long src = ((long long)o * T + tap) * C + c;
      split1(W[src], h, l);
      if (hi) { hi[src] = h; lo[src] = l; }
    }
    th[ty + 8 * i][tx] = h; tl[ty + 8 * i][tx] = l;
  }
  if (!hiT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, o = o0 + tx;
    if (o < O && c < C) {
      const long long dst = ((long long)c * T + tap) * O + o;
      hiT[dst] = th[tx][ty + 8 * i];
      loT[dst] = tl[tx][ty + 8 * i];
    }
  }
}

}  // namespace

bool cape_gemm_ws_eligible(const GemmP& p, int a_mode) {
  static const bool off = getenv("CAPE_GEMM_NO_WS") != nullptr;    // tuning switch: always use the general kernel
  if (off || !p.Bhi || !p.Blo || p.split_k != 1 || p.colsum_out || p.mask_src) return false;
  if (p.K % BK != 0 || p.K <= 0) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(p.A) || !al16(p.Bhi) || !al16(p.Blo) || p.ldp % 8 != 0) return false;
  if (a_mode == 0) return p.lda % 4 == 0;
  if (a_mode == 2) return p.cC % 16 == 0;
  if (a_mode == 3) return p.cO % 16 == 0;
  return false;
}

int cape_gemm_ws_launch(GemmP& p, int a_mode, hipStream_t stream) {
  // 128-row blocks (4 waves) when that still yields >= 2 blocks per CU, else 64-row blocks (2 waves) for fill
  const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + WS_BN - 1) / WS_BN);
  static const char* force = getenv("CAPE_GEMM_WS_ROWS");          // tuning override: 64 or 128
  bool big = t128 >= 512;
  if (force) big = force[0] == '1';
  const int BM = big ? 128 : 64;
  p.tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.N + WS_BN - 1) / WS_BN;
  const long long ntiles = (long long)p.tilesM * p.tilesN;
  if (ntiles >= (1ll << 31)) return cape_set_error("cape_gemm_f32: too many tiles");
  const dim3 grid((unsigned)ntiles);
#define WS_CASE(AM)                                                                                              \
  if (a_mode == AM) {                                                                                            \
    if (big) hipLaunchKernelGGL((gemm_ws_kernel<4, AM>), grid, dim3(256), 0, stream, p);                         \
    else hipLaunchKernelGGL((gemm_ws_kernel<2, AM>), grid, dim3(128), 0, stream, p);                             \
  }
  WS_CASE(0) WS_CASE(2) WS_CASE(3)
#undef WS_CASE
  CAPE_LAUNCH_CHECK("cape_gemm_f32(ws)");
  return 0;
}

extern "C" int cape_split_planes(const float* W, int O, int T, int C, uint16_t* hi, uint16_t* lo, uint16_t* hiT, uint16_t* loT,
                                 cape_stream_t stream) {
  CAPE_REQUIRE(W != nullptr && O > 0 && T > 0 && C > 0, "cape_split_planes: bad arguments");
  CAPE_REQUIRE((hi != nullptr) == (lo != nullptr) && (hiT != nullptr) == (loT != nullptr) && (hi || hiT),
               "cape_split_planes: planes come in (hi, lo) pairs and at least one pair is needed");
  CAPE_REQUIRE(T <= 65535 && (O + 31) / 32 <= 65535, "cape_split_planes: shape too large");
  hipLaunchKernelGGL(split_planes_kernel, dim3((C + 31) / 32, (O + 31) / 32, T), dim3(256), 0, as_stream(stream), W, O, T, C, hi,
                     lo, hiT, loT);
  CAPE_LAUNCH_CHECK("cape_split_planes");
  return 0;
}
