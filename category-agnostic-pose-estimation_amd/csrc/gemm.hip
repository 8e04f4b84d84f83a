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
#include "gemm_tile.h"

namespace {

template <int BM, int BN, int AMODE, int BMODE, bool VEC, int PREC, bool KFULL>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmP pin) {
  GemmP p = pin;
  if (gridDim.y > 1) {                                             // batched launch: uniform operand offsets
    const int b0 = blockIdx.y / p.bdiv, b1 = blockIdx.y - b0 * p.bdiv;
    p.A += b0 * p.sA0 + b1 * p.sA1;
    p.B += b0 * p.sB0 + b1 * p.sB1;
    p.C += b0 * p.sC0 + b1 * p.sC1;
    if (p.bias) p.bias += b0 * p.sBias0 + b1 * p.sBias1;
  }
  gemm_tile_body<BM, BN, AMODE, BMODE, VEC, PREC, KFULL>(p, (int)blockIdx.x);
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
    if (p.residual && (p.res_cols == 0 || n < p.res_cols)) v += p.residual[(long long)row * p.ldr + n];
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
  p.res_cols = d->res_cols; p.sBias0 = d->sBias0; p.sBias1 = d->sBias1;
  if (d->cKHp > 0) {
    CAPE_REQUIRE(d->a_mode == 3 && d->b_mode == 2 && d->cStride == 1 && d->cO % 32 == 0 && d->K % 32 == 0 && d->cKWp > 0 &&
                     d->cTapHS >= 1 && d->cTapWS >= 1 && d->cTapH0 >= 0 && d->cTapW0 >= 0 &&
                     d->cTapH0 + (d->cKH - 1) * d->cTapHS < d->cKHp && d->cTapW0 + (d->cKW - 1) * d->cTapWS < d->cKWp,
                 "cape_gemm_f32: tap sub-lattice needs conv-dgrad modes, stride 1, O %% 32 == 0 and taps inside the physical filter");
    p.cPadX = d->cPadX; p.cKHp = d->cKHp; p.cKWp = d->cKWp;
    p.cTapH0 = d->cTapH0; p.cTapHS = d->cTapHS; p.cTapW0 = d->cTapW0; p.cTapWS = d->cTapWS;
  } else {
    p.cPadX = d->cPad; p.cKHp = d->cKH; p.cKWp = d->cKW; p.cTapH0 = 0; p.cTapHS = 1; p.cTapW0 = 0; p.cTapWS = 1;
  }
  CAPE_REQUIRE(d->res_cols >= 0 && d->res_cols % 32 == 0, "cape_gemm_f32: res_cols must be a non-negative multiple of 32");
  if (d->batch > 1)
    CAPE_REQUIRE(d->batch <= 65535 && d->split_k == 1 && !d->scale && !d->residual && !d->mask_src && !d->colsum_out &&
                     d->dropout_p == 0.f && (d->a_mode == 0 || d->a_mode == 1) && (d->b_mode == 0 || d->b_mode == 1),
                 "cape_gemm_f32: batched launches take dense modes, no epilogue vector besides the bias, split_k 1, batch <= 65535");
  if (d->mask_src) CAPE_REQUIRE(d->split_k == 1, "cape_gemm_f32: mask_src needs split_k == 1");
  CAPE_REQUIRE(d->precision >= 0 && d->precision <= 2, "cape_gemm_f32: precision must be 0 (fp32), 1 (bf16x3) or 2 (single bf16)");
  p.single = d->precision == 2;
  const int prec = d->precision == 0 ? 0 : 1;                       // kernel family: exact fp32 MFMA, or the bf16 split (3 or 1 MFMA per product)
  // register-stationary weights (gemm_rs.hip): dense A against a <= 256-deep weight, the token products of the transformer
  if (prec == 1 && d->batch <= 1 && cape_gemm_rs_eligible(p, d->a_mode, d->b_mode)) return cape_gemm_rs_launch(p, d->b_mode, as_stream(stream));

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
  bool big = d->K / d->split_k >= 1024 && t128 >= 256 && (long long)d->M * d->N >= 256 * 1024 && prec == 1 && vec;
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
  int rc = big ? launch_mode<128, 128>(p, d->a_mode, d->b_mode, vec, prec, grid, as_stream(stream))
               : launch_mode<64, 64>(p, d->a_mode, d->b_mode, vec, prec, grid, as_stream(stream));
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
