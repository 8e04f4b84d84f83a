// gemm_group.hip -- grouped launch of the GEMM family: up to CAPE_GEMM_GROUP_MAX independent products in ONE launch.
//
// Why (round 2 numbers, profiles/r02_bench_kernel_stats.csv): the weight gradients are 199 of the ~700 GEMM launches of a
// training step and a quarter of its kernel time.  Each is a small output (256 x 256 ... 1024 x 256) over a long reduction
// (the token rows), so a launch of its own has to cut K into 24-64 splits to cover the chip, and every split pays a whole
// tile of memory-side float atomics (16.7 MB for one 256 x 256 x 43520 product, ~13 us at the 1.3 TB/s those atomics run
// at) besides the launch itself.  Nothing downstream waits for a weight gradient until the optimizer step: the host side
// (hip/functional.py, Runtime.defer_wgrad) therefore *queues* these products during the backward pass and hands 8-32 of them
// to this entry point at a time.  The tiles of all products fill the chip together, so each needs 4-8x fewer k-splits
// (atomic volume and launch count both drop), and one descriptor table travels in the kernel arguments (no device table,
// nothing to keep alive, capturable into a hipGraph as it is).
//
// The tile body is the family's (gemm_tile.h): same LDS images, same bf16x3 / fp32 arithmetic, same fused bias-gradient
// column sums.  Only the weight-gradient operand modes are instantiated: A = dY^T (a_mode 1) against the activation stored
// [tokens][K] (b_mode 1) or gathered as im2col rows (b_mode 3).
#include "gemm_tile.h"

namespace {

constexpr int GROUP_MAX = CAPE_GEMM_GROUP_MAX;

struct GroupItem {                                  // 112 bytes
  const float* A; const float* B; float* C; float* colsum_out;
  int M, N, K, split_k;
  int lda, ldb, ldc, tilesM;
  int tilesN, cN, cH, cW;
  int cC, cKH, cKW, cStride;
  int cPad, cOH, cOW, cO;
};

struct GroupArgs {
  int n, single;
  int bstart[GROUP_MAX + 1];                        // first block of item i (multiples of 8); bstart[n] = grid size
  GroupItem it[GROUP_MAX];
};
static_assert(sizeof(GroupArgs) <= 4096, "the item table travels as kernel arguments");

template <int BM, int BN, int BMODE, int PREC, bool KFULL>
__global__ void __launch_bounds__(256) gemm_group_kernel(const GroupArgs g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.bstart[i + 1]) ++i;  // uniform: a short scalar scan of the kernel-argument table
  const GroupItem& q = g.it[i];
  GemmP p;
  p.M = q.M; p.N = q.N; p.K = q.K;
  p.A = q.A; p.lda = q.lda; p.B = q.B; p.ldb = q.ldb; p.C = q.C; p.ldc = q.ldc;
  p.cN = q.cN; p.cH = q.cH; p.cW = q.cW; p.cC = q.cC; p.cKH = q.cKH; p.cKW = q.cKW;
  p.cStride = q.cStride; p.cPad = q.cPad; p.cOH = q.cOH; p.cOW = q.cOW; p.cO = q.cO;
  p.cPadX = q.cPad; p.cKHp = q.cKH; p.cKWp = q.cKW; p.cTapH0 = 0; p.cTapHS = 1; p.cTapW0 = 0; p.cTapWS = 1;
  p.scale = nullptr; p.bias = nullptr; p.residual = nullptr; p.ldr = 0;
  p.relu = 0; p.accumulate = 1; p.split_k = q.split_k;
  p.colsum_out = q.colsum_out;
  p.drop_thresh = 0; p.inv_keep = 1.f; p.rng_state = nullptr; p.rng_stream = 0;
  p.tilesM = q.tilesM; p.tilesN = q.tilesN;
  p.Bpack = nullptr; p.mask_src = nullptr; p.ldm = 0; p.mask_scale = 1.f;
  p.bdiv = 1; p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = 0;
  p.res_cols = 0; p.sBias0 = p.sBias1 = 0; p.single = g.single;
  gemm_tile_body<BM, BN, 1, BMODE, true, PREC, KFULL>(p, b - g.bstart[i]);
}

template <int BM, int BMODE>
void launch_group(const GroupArgs& g, int prec, bool kfull, hipStream_t s) {
  const dim3 grid((unsigned)g.bstart[g.n]), block(256);
  if (prec == 1) {
    if (kfull) hipLaunchKernelGGL((gemm_group_kernel<BM, BM, BMODE, 1, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_group_kernel<BM, BM, BMODE, 1, false>), grid, block, 0, s, g);
  } else {
    if (kfull) hipLaunchKernelGGL((gemm_group_kernel<BM, BM, BMODE, 0, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_group_kernel<BM, BM, BMODE, 0, false>), grid, block, 0, s, g);
  }
}

}  // namespace

extern "C" int cape_gemm_group_f32(const cape_gemm_desc* descs, int n, int tile, cape_stream_t stream) {
  CAPE_REQUIRE(descs != nullptr && n >= 1 && n <= GROUP_MAX, "cape_gemm_group_f32: n=%d must be in 1..%d", n, GROUP_MAX);
  CAPE_REQUIRE(tile == 64 || tile == 128, "cape_gemm_group_f32: tile must be 64 or 128");
  const int b_mode = descs[0].b_mode, prec_in = descs[0].precision, prec = prec_in == 0 ? 0 : 1;
  CAPE_REQUIRE(b_mode == 1 || b_mode == 3, "cape_gemm_group_f32: only the weight-gradient modes (a_mode 1, b_mode 1 or 3) are grouped");
  CAPE_REQUIRE(prec_in >= 0 && prec_in <= 2, "cape_gemm_group_f32: precision must be 0 (fp32), 1 (bf16x3) or 2 (single bf16)");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  GroupArgs g;
  g.n = n;
  g.single = prec_in == 2;
  bool kfull = true;
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    const cape_gemm_desc& d = descs[i];
    CAPE_REQUIRE(d.a_mode == 1 && d.b_mode == b_mode && d.precision == prec_in, "cape_gemm_group_f32: item %d: mixed modes / precisions", i);
    CAPE_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.A && d.B && d.C, "cape_gemm_group_f32: item %d: empty product or null operand", i);
    CAPE_REQUIRE(d.accumulate && !d.scale && !d.bias && !d.residual && !d.relu && d.dropout_p == 0.f && !d.mask_src && d.batch <= 1,
                 "cape_gemm_group_f32: item %d: grouped products accumulate onto C and take no other epilogue operand", i);
    CAPE_REQUIRE(d.split_k >= 1, "cape_gemm_group_f32: item %d: split_k must be >= 1", i);
    // the vector path of the tile body: aligned bases, 16-byte rows
    CAPE_REQUIRE(al16(d.A) && al16(d.B) && d.lda % 4 == 0 && d.M % 4 == 0 && d.M >= 4 && d.N % 4 == 0 && d.N >= 4 &&
                     (b_mode == 3 || d.ldb % 4 == 0),
                 "cape_gemm_group_f32: item %d: operands must be 16-byte aligned with M, N, lda, ldb multiples of 4", i);
    CAPE_REQUIRE(d.lda < (1ll << 31) && d.ldb < (1ll << 31) && d.ldc < (1ll << 31), "cape_gemm_group_f32: item %d: leading dimension too large", i);
    if (b_mode == 3) {
      const long long taps = (long long)d.cKH * d.cKW;
      CAPE_REQUIRE(d.cC % 4 == 0 && d.cO % 4 == 0 && d.cStride >= 1 && d.cKH >= 1 && d.cKW >= 1, "cape_gemm_group_f32: item %d: bad conv geometry", i);
      CAPE_REQUIRE(d.N == taps * d.cC && d.K == (long long)d.cN * d.cOH * d.cOW && d.M == d.cO, "cape_gemm_group_f32: item %d: conv-wgrad shape mismatch", i);
    }
    GroupItem& q = g.it[i];
    q.A = d.A; q.B = d.B; q.C = d.C; q.colsum_out = d.colsum_out;
    q.M = d.M; q.N = d.N; q.K = d.K; q.split_k = d.split_k;
    q.lda = (int)d.lda; q.ldb = (int)d.ldb; q.ldc = (int)d.ldc;
    q.tilesM = (d.M + tile - 1) / tile; q.tilesN = (d.N + tile - 1) / tile;
    q.cN = d.cN; q.cH = d.cH; q.cW = d.cW; q.cC = d.cC; q.cKH = d.cKH; q.cKW = d.cKW;
    q.cStride = d.cStride; q.cPad = d.cPad; q.cOH = d.cOH; q.cOW = d.cOW; q.cO = d.cO;
    kfull = kfull && (d.K % BK == 0);
    g.bstart[i] = (int)blocks;
    const long long nb = (long long)q.tilesM * q.tilesN * d.split_k;
    blocks += (nb + 7) / 8 * 8;
    CAPE_REQUIRE(blocks < (1ll << 30), "cape_gemm_group_f32: grid too large");
  }
  g.bstart[n] = (int)blocks;
  for (int i = n + 1; i <= GROUP_MAX; ++i) g.bstart[i] = (int)blocks;
  hipStream_t s = as_stream(stream);
  if (tile == 64) { if (b_mode == 1) launch_group<64, 1>(g, prec, kfull, s); else launch_group<64, 3>(g, prec, kfull, s); }
  else { if (b_mode == 1) launch_group<128, 1>(g, prec, kfull, s); else launch_group<128, 3>(g, prec, kfull, s); }
  CAPE_LAUNCH_CHECK("cape_gemm_group_f32");
  return 0;
}
