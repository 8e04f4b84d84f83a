// gemm_rs.hip -- "register-stationary weights" member of the GEMM family (bf16x3 split, gfx950).
//
//   C[M,N] = epi(A[M,K] * W),  A dense fp32 [M][K] (a_mode 0),  W = B stored [N][K] (b_mode 0) or [K][N] (b_mode 1),
//   K in {64, 128, 256}: the token products of the transformer (M = 43520 / 6400 rows against a 256-deep weight) and the
//   narrow 1x1 convolutions of the trunk.
//
// Why a second kernel (measured on MI355X, round 1): the tiled kernel of gemm.hip moves a 64-wide output tile per block, so an
// M x 256 x 256 product re-reads A four times and the 256 KB weight 680 times through L2 (400 MB for 89 MB of operands) and
// pays eight k-tiles of load->LDS->barrier latency per tile: 39 us against a 14 us HBM time.  Here the *weight* never moves:
//   * a block is 8 waves (512 threads, 2 waves per SIMD); wave w owns 32 output columns and keeps their whole K-deep weight
//     column block as MFMA B-fragments in registers, already split into bf16 (hi, lo) planes: K/16 steps x 8 VGPRs
//     (128 VGPRs at K = 256), loaded and split once per block;
//   * blocks are persistent over row units (64 rows): A is streamed exactly once per column chunk -- global -> registers
//     (one unit ahead) -> split to bf16 (hi, lo) -> LDS image [row][k] (528-byte row stride: conflict-free ds_read_b128) ->
//     fragments; one barrier per 64 rows, LDS double-buffered; all 8 waves share the converted unit (the split costs each
//     thread 4-8 float4 per unit against 48-96 MFMAs);
//   * NW = 8: the block covers 256 columns (each wave both 32-row halves of the unit); NW = 4: 128 columns, waves 0-3 / 4-7
//     take one half each (N = 384, or few rows: twice the blocks);
//   * column chunks of one row unit run on the same XCD (block id = xcd + 8 * (chunk + nchunks * slot)), so the unit is
//     fetched from HBM once and re-read from that L2;
//   * epilogue identical to gemm_kernel's (scale, bias, residual, relu, dropout, gate, accumulate), same dropout counter
//     indexing, so the two kernels are interchangeable bit-for-bit in masks (not in summation order).
#include <stdlib.h>
#include "gemm_common.h"

namespace {

constexpr int RS_UNIT = 64;                       // rows per unit

#ifdef RS_STAMPS                                  // lab builds only (tools/lab/rs_lab.hip): per-block phase time stamps
__device__ long long* g_rs_stamps;
#define RS_STAMP(i) do { if (threadIdx.x == 0 && (i) < 16) g_rs_stamps[blockIdx.x * 16 + (i)] = clock64(); } while (0)
#define RS_FINE(cond, i) do { if ((cond) && threadIdx.x == 0) g_rs_stamps[blockIdx.x * 16 + 8 + (i)] = clock64(); } while (0)
#else
#define RS_STAMP(i) do {} while (0)
#define RS_FINE(cond, i) do {} while (0)
#endif

template <int KS>
struct RsGeom {
  static constexpr int K = 16 * KS;
  static constexpr int LD = K + 8;                // bf16 elements per LDS row (16-byte pad)
  static constexpr int PLANE = RS_UNIT * LD;      // elements per plane
  static constexpr int NV = KS / 2;               // float4 per thread per unit (64 rows * K / 4 / 512)
  static constexpr size_t LDS_BYTES = (size_t)2 /*buffers*/ * 2 /*planes*/ * PLANE * 2;
};

// EPI: 0 scale/bias/relu only; 1 + residual; 2 + gate (mask_src); 3 accumulate onto C; 4 + dropout
template <int NW, int BMODE, int KS, int EPI>
__global__ void __launch_bounds__(512) gemm_rs_kernel(const GemmP p, int nchunks, int gpc) {
  using G = RsGeom<KS>;
  constexpr int K = G::K, LD = G::LD, NV = G::NV;
  constexpr int CPR = K / 4;                      // float4 chunks per row
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % NW, wm = wave / NW;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, loc = bid >> 3;
  const int chunk = loc % nchunks;
  const int slot = (loc / nchunks) * 8 + xcd;      // 0 .. gpc-1: this block's position among the blocks of its chunk
  const int n0 = chunk * (32 * NW) + 32 * wn;
  const int nunits = (p.M + RS_UNIT - 1) / RS_UNIT;
  if (slot >= nunits) return;

  // ---- staging: 64 rows x K floats of a row-major matrix, global -> registers -> bf16 (hi, lo) LDS image [row][k]
  const int st_row = t / CPR, st_kc = t % CPR;     // chunk i of this thread: row st_row + (512 / CPR) * i, k = 4 * st_kc
  constexpr int RSTEP = 512 / CPR;
  auto load_rows = [&](float4 (&dst)[NV], const float* base, long long ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = min(row0 + st_row + RSTEP * i, nrows - 1);
      dst[i] = *reinterpret_cast<const float4*>(base + (long long)row * ld + 4 * st_kc);
    }
  };
  auto store_rows = [&](float4 (&src)[NV], int buf) {
    unsigned short* Ph = lds + buf * 2 * G::PLANE;
    unsigned short* Pl = Ph + G::PLANE;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int o = (st_row + RSTEP * i) * LD + 4 * st_kc;
      unsigned h0, l0, h1, l1;
      split2(src[i].x, src[i].y, h0, l0);
      split2(src[i].z, src[i].w, h1, l1);
      *reinterpret_cast<uint2*>(Ph + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(Pl + o) = make_uint2(l0, l1);
    }
  };

  // ---- epilogue constants of this lane's column (loaded first: every later wait then covers them)
  const int col = n0 + r;
  const bool col_ok = col < p.N;
  const float sc = p.scale ? p.scale[min(col, p.N - 1)] : 1.f;
  const float bi = p.bias ? p.bias[min(col, p.N - 1)] : 0.f;
  const float floor_v = p.relu ? 0.f : -__builtin_inff();      // relu as one v_max
  uint64_t seed = 0, step = 0;
  if constexpr (EPI == 4) { seed = p.rng_state[0]; step = p.rng_state[1]; }
  // 32-bit element offsets from a wave-uniform row base (host-checked: every M * ld < 2^31)
  const int ldc = (int)p.ldc;
  const unsigned lo_c = (unsigned)(4 * h * ldc + col);
  // EPI 1: the residual has a leading dimension of its own (q | k | v as one product, `+ query_pos` on the q columns: ldr = 256
  // against ldc = 768) and may end at res_cols -- wave-uniform, a wave's 32 columns lie on one side of the limit
  const int ldx = EPI == 1 ? (int)p.ldr : ldc;
  const unsigned lo_x = (unsigned)(4 * h * ldx + col);
  const bool res_on = EPI != 1 || !p.res_cols || n0 < p.res_cols;

  auto store_piece = [&](float4& v, int i, int buf) {          // one float4 of a unit: split + two 8-byte LDS stores
    unsigned short* Ph = lds + buf * 2 * G::PLANE + (st_row + RSTEP * i) * LD + 4 * st_kc;
    unsigned h0, l0, h1, l1;
    split2(v.x, v.y, h0, l0);
    split2(v.z, v.w, h1, l1);
    *reinterpret_cast<uint2*>(Ph) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(Ph + G::PLANE) = make_uint2(l0, l1);
  };

  RS_STAMP(0);
  // ---- this wave's weight column block as B fragments: lane (r, h) holds W[k = 16 s + 8 h + j][n0 + r], j = 0..7
  bf16x8 bhi[KS], blo[KS];
  float4 ra[NV];                                   // A staging registers: a unit lives here between its load and its LDS store
  if (p.Bpack) {
    // fragment-ordered bf16 planes prepared once per optimizer step (cape_pack_weights): block (column group, k-step, plane)
    // = 64 lanes x 16 bytes, so the whole weight block of this wave arrives as 2 * K/16 coalesced 1 KB loads -- no LDS pass,
    // no split arithmetic, no barrier before the first MFMA
    load_rows(ra, p.A, p.lda, slot * RS_UNIT, p.M);
    // (a wave whose 32 columns lie entirely beyond N reads the last packed group: valid memory, results never stored)
    const int grp = min(n0 >> 5, (p.N + 31) / 32 - 1);
    const uint4* bp = reinterpret_cast<const uint4*>(p.Bpack) + ((long long)grp * KS * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bhi[s] = __builtin_bit_cast(bf16x8, bp[(2 * s + 0) * 64]);
      blo[s] = __builtin_bit_cast(bf16x8, bp[(2 * s + 1) * 64]);
    }
  } else if constexpr (BMODE == 0) {
    // W stored [N][K]: the same coalesced row staging as A, 64 weight rows (two waves' columns) per pass through the two
    // LDS buffers; the owning waves then read their fragments with the conflict-free pattern of the A reads
    constexpr int NPASS = NW / 2;
    const int nb = chunk * (32 * NW);
    float4 rw[2][NV];
    load_rows(rw[0], p.B, p.ldb, nb, p.N);
    if (NPASS > 1) load_rows(rw[1], p.B, p.ldb, nb + 64, p.N);
    load_rows(ra, p.A, p.lda, slot * RS_UNIT, p.M);
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      store_rows(rw[q & 1], q & 1);
      if (q + 2 < NPASS) load_rows(rw[q & 1], p.B, p.ldb, nb + 64 * (q + 2), p.N);
      __syncthreads();
      if ((wn >> 1) == q) {
        const unsigned short* Ph = lds + (q & 1) * 2 * G::PLANE + (32 * (wn & 1) + r) * LD + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          bhi[s] = *reinterpret_cast<const bf16x8*>(Ph + 16 * s);
          blo[s] = *reinterpret_cast<const bf16x8*>(Ph + G::PLANE + 16 * s);
        }
      }
    }
    __syncthreads();                               // the last pass has been read: buffer 0 / 1 may take A units
  } else {
    // W stored [K][N]: 32 consecutive columns per k row = one full 128-byte line per half wave and load instruction
    load_rows(ra, p.A, p.lda, slot * RS_UNIT, p.M);
    const int n = min(n0 + r, p.N - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float f[8];
      const int k = 16 * s + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = p.B[(long long)(k + j) * p.ldb + n];
      unsigned hw[4], lw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2(f[2 * j], f[2 * j + 1], hw[j], lw[j]);
      bhi[s] = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
      blo[s] = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
    }
  }

  RS_STAMP(1);
  store_rows(ra, 0);
  if (slot + gpc < nunits) load_rows(ra, p.A, p.lda, (slot + gpc) * RS_UNIT, p.M);
  __syncthreads();
  RS_STAMP(2);
  int stamp_i = 3;

  // Order inside an iteration (all waves): MFMAs of unit u; the split + LDS store of unit u+1 (whose loads were issued
  // a whole iteration earlier) rides in the last K/32 k-steps of the last half, one float4 per k-step, in the shadow of the
  // dependent MFMA chain; stores of the results; then the loads of unit u+2 are issued; barrier.  vmcnt retires in issue
  // order, so a wait for a load also waits for every older store: here the only stores older than the awaited loads are
  // those of the previous iteration, the younger ones of this iteration are counted exactly by the compiler (straight-line
  // code), and no load is pending behind a store across the loop edge.
  int buf = 0;
  for (int u = slot; u < nunits; u += gpc) {
    const bool has_next = u + gpc < nunits;
    const unsigned short* Ah = lds + buf * 2 * G::PLANE;
    const unsigned short* Al = Ah + G::PLANE;
    const bool full = (u + 1) * RS_UNIT <= p.M;
    constexpr int H1 = (NW == 8) ? 2 : 1;
    // EPI 1..3: the extra operand (residual / gate source / old C) of a tile is requested before its first
    // MFMA: left in the epilogue, its round trip sat between the last MFMA and the stores of every tile (the gated FFN
    // dgrad ran 2x longer than the plain product of the same shape)
#pragma unroll
    for (int hh = 0; hh < H1; ++hh) {
      const int half = (NW == 8) ? hh : wm;
      float xv[16];
      if constexpr (EPI >= 1 && EPI <= 3) {
        if (col_ok && res_on) {
          const int rb = u * RS_UNIT + 32 * half;
          // (gate source / old C share the 32-bit lane offsets of the C stores (eligibility: ldm == ldc), only the SGPR base
          // differs; the residual brings one lane offset of its own)
          const float* xb = EPI == 1 ? p.residual : EPI == 2 ? p.mask_src : p.C;
          const float* x0 = xb + (long long)rb * ldx;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int roff = (i & 3) + 8 * (i >> 2);
            if (full) xv[i] = x0[lo_x + (unsigned)(roff * ldx)];
            else xv[i] = xb[(long long)min(rb + roff + 4 * h, p.M - 1) * ldx + col];   // ragged last unit: clamped rows, never stored
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) xv[i] = 0.f;
        }
      }
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      RS_FINE(u == slot + gpc, 3 * hh);
      const int ao = (32 * half + r) * LD + 8 * h;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(Ah + ao + 16 * s);
        const bf16x8 alo = *reinterpret_cast<const bf16x8*>(Al + ao + 16 * s);
        if (!p.single) {                                             // uniform; precision 2 keeps hi x hi only
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[s], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[s], acc, 0, 0, 0);
        if (hh == H1 - 1 && s >= KS - NV && has_next) store_piece(ra[s - (KS - NV)], s - (KS - NV), buf ^ 1);
      }
      RS_FINE(u == slot + gpc, 3 * hh + 1);
      // C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).  The row base of
      // register i is wave-uniform (scalar address arithmetic); the lane part is a 32-bit offset.
      if (!col_ok) continue;
      const int rb = u * RS_UNIT + 32 * half;                   // uniform
      const int rlim = full ? 64 : p.M - rb - 4 * h;            // register i of this lane is a valid row iff roff(i) < rlim
      float* c0 = p.C + (long long)rb * ldc;
      if constexpr (EPI == 0) {
        if (full) {
#pragma unroll
          for (int i = 0; i < 16; ++i) c0[lo_c + (unsigned)(((i & 3) + 8 * (i >> 2)) * ldc)] = fmaxf(fmaf(acc[i], sc, bi), floor_v);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int roff = (i & 3) + 8 * (i >> 2);
            if (roff < rlim) c0[lo_c + (unsigned)(roff * ldc)] = fmaxf(fmaf(acc[i], sc, bi), floor_v);
          }
        }
      } else if constexpr (EPI == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int roff = (i & 3) + 8 * (i >> 2);
          const int row = rb + roff + 4 * h;
          float v = fmaxf(fmaf(acc[i], sc, bi), floor_v);
          v = cape_keep(seed, step, p.rng_stream, (uint64_t)row * (uint64_t)p.N + col, p.drop_thresh) ? v * p.inv_keep : 0.f;
          if (roff < rlim) c0[lo_c + (unsigned)(roff * ldc)] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int roff = (i & 3) + 8 * (i >> 2);
          const float x = xv[i];
          float v = fmaf(acc[i], sc, bi);
          if constexpr (EPI == 1) v = fmaxf(v + x, floor_v);
          if constexpr (EPI == 2) v = x != 0.f ? fmaxf(v, floor_v) * p.mask_scale : 0.f;
          if constexpr (EPI == 3) v = fmaxf(v, floor_v) + x;
          if (roff < rlim) c0[lo_c + (unsigned)(roff * ldc)] = v;
        }
      }
    }
    RS_FINE(u == slot + gpc, 6);
    if (u + 2 * gpc < nunits) load_rows(ra, p.A, p.lda, (u + 2 * gpc) * RS_UNIT, p.M);
    __syncthreads();
    RS_FINE(u == slot + gpc, 7);
    RS_STAMP(stamp_i); ++stamp_i;
    buf ^= 1;
  }
}

static int rs_epi(const GemmP& p) {              // which single extra epilogue operand a launch carries; -1: more than one
  const int n = (p.residual != nullptr) + (p.mask_src != nullptr) + (p.accumulate != 0) + (p.drop_thresh != 0);
  if (n == 0) return 0;
  if (n > 1) return -1;
  return p.residual ? 1 : p.mask_src ? 2 : p.accumulate ? 3 : 4;
}

template <int NW, int BMODE, int KS, int EPI>
int rs_launch_epi(const GemmP& p, int nchunks, int gpc, hipStream_t s) {
  static bool attr_set = false;                    // > 64 KB of LDS needs the opt-in once per kernel
  auto kfn = gemm_rs_kernel<NW, BMODE, KS, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)RsGeom<KS>::LDS_BYTES);
    if (e != hipSuccess) return cape_set_error("cape_gemm_f32(rs): hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)(gpc * nchunks)), dim3(512), RsGeom<KS>::LDS_BYTES, s, p, nchunks, gpc);
  return 0;
}

template <int NW, int BMODE, int KS>
int rs_launch(const GemmP& p, int nchunks, int gpc, hipStream_t s) {
  switch (rs_epi(p)) {
    case 0: return rs_launch_epi<NW, BMODE, KS, 0>(p, nchunks, gpc, s);
    case 1: return rs_launch_epi<NW, BMODE, KS, 1>(p, nchunks, gpc, s);
    case 2: return rs_launch_epi<NW, BMODE, KS, 2>(p, nchunks, gpc, s);
    case 3: return rs_launch_epi<NW, BMODE, KS, 3>(p, nchunks, gpc, s);
    case 4: return rs_launch_epi<NW, BMODE, KS, 4>(p, nchunks, gpc, s);
  }
  return cape_set_error("cape_gemm_f32(rs): unsupported epilogue combination");
}

// ---- weight packing: W (as the B operand of mode b_mode) -> fragment-ordered bf16 (hi, lo) planes -------------------------
struct PackItem { const float* B; unsigned short* out; long long ldb; int N, K, b_mode, pad; };
static_assert(sizeof(PackItem) == sizeof(cape_pack_item), "cape_pack_item layout");

// one wave per (column group, k-step): lane (r, h) converts W_eff[k = 16 s + 8 h + j][n = 32 g + r], j = 0..7
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackItem* items, int n_items) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int it = blockIdx.y; it < n_items; it += gridDim.y) {
    const PackItem q = items[it];
    const int KS = q.K >> 4, groups = (q.N + 31) >> 5;
    for (int w = blockIdx.x * 4 + wave; w < groups * KS; w += gridDim.x * 4) {
      const int g = w / KS, s = w - g * KS;
      const int n = 32 * g + r, k = 16 * s + 8 * h;
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      if (n < q.N) {
        if (q.b_mode == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = q.B[(long long)n * q.ldb + k + j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = q.B[(long long)(k + j) * q.ldb + n];
        }
      }
      unsigned hw[4], lw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2(f[2 * j], f[2 * j + 1], hw[j], lw[j]);
      uint4* o = reinterpret_cast<uint4*>(q.out) + ((long long)w * 2) * 64 + lane;
      o[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      o[64] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
}

}  // namespace

extern "C" size_t cape_packed_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (size_t)((N + 31) / 32) * 32 * (size_t)K * 4;          // two bf16 planes, column groups padded to 32
}

extern "C" int cape_pack_weights(const cape_pack_item* items_dev, int n_items, int max_blocks_per_item, cape_stream_t stream) {
  CAPE_REQUIRE(items_dev != nullptr && n_items >= 0 && max_blocks_per_item >= 1, "cape_pack_weights: bad arguments");
  if (n_items == 0) return 0;
  const unsigned gy = (unsigned)(n_items < 65535 ? n_items : 65535);
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)max_blocks_per_item, gy), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const PackItem*>(items_dev), n_items);
  CAPE_LAUNCH_CHECK("cape_pack_weights");
  return 0;
}

bool cape_gemm_rs_eligible(const GemmP& p, int a_mode, int b_mode) {
  static const bool off = getenv("CAPE_GEMM_NO_RS") != nullptr;      // tuning switch: always use the tiled kernel
  if (off || a_mode != 0 || (b_mode != 0 && b_mode != 1) || p.split_k != 1 || p.colsum_out) return false;
  if (p.K != 64 && p.K != 128 && p.K != 256) return false;
  if (p.M <= 64 || p.N < 32 || rs_epi(p) < 0) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(p.A) || p.lda % 4 != 0) return false;
  if (p.Bpack && !al16(p.Bpack)) return false;
  if (!p.Bpack && b_mode == 0 && (!al16(p.B) || p.ldb % 4 != 0)) return false;
  const long long lim = 1ll << 31;                                   // 32-bit lane offsets in the epilogue
  if ((long long)p.M * p.ldc >= lim) return false;
  if (p.mask_src && p.ldm != p.ldc) return false;                    // the gate source shares C's lane offsets
  if (p.residual && (long long)p.M * p.ldr >= lim) return false;

  return true;
}

int cape_gemm_rs_launch(const GemmP& p, int b_mode, hipStream_t stream) {
  const int nunits = (p.M + RS_UNIT - 1) / RS_UNIT;
  // 256-column blocks when N fills them and there are enough row units to occupy the chip; else 128-column blocks
  static const char* force = getenv("CAPE_GEMM_RS_NW");              // tuning override: 4 or 8
  bool wide = (p.N % 256 == 0) && (long long)nunits * (p.N / 256) >= 192;
  if (force) wide = force[0] == '8';
  const int NW = wide ? 8 : 4;
  const int nchunks = (p.N + 32 * NW - 1) / (32 * NW);
  int gpc = (256 / nchunks) / 8 * 8;
  if (gpc < 8) gpc = 8;
  const int need = (nunits + 7) / 8 * 8;
  if (gpc > need) gpc = need;
  if ((long long)gpc * nchunks >= (1ll << 31)) return cape_set_error("cape_gemm_f32(rs): grid too large");
  int rc = 0;
#define RS_CASE(NW_, BM_, KS_)                                                                 \
  if (NW == NW_ && b_mode == BM_ && p.K == 16 * KS_) rc = rs_launch<NW_, BM_, KS_>(p, nchunks, gpc, stream);
  RS_CASE(8, 0, 16) RS_CASE(8, 0, 8) RS_CASE(8, 0, 4) RS_CASE(8, 1, 16) RS_CASE(8, 1, 8) RS_CASE(8, 1, 4)
  RS_CASE(4, 0, 16) RS_CASE(4, 0, 8) RS_CASE(4, 0, 4) RS_CASE(4, 1, 16) RS_CASE(4, 1, 8) RS_CASE(4, 1, 4)
#undef RS_CASE
  if (rc) return rc;
  CAPE_LAUNCH_CHECK("cape_gemm_f32(rs)");
  return 0;
}
