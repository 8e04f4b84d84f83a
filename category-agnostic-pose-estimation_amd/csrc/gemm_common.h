// gemm_common.h -- pieces shared by the GEMM translation units (gemm.hip, gemm_ws.hip).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// PREC 1 = "bf16x3": every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits)
// when it is staged into LDS, and each 16-deep k-step is three bf16 MFMAs hi*hi + hi*lo + lo*hi accumulated in fp32
// (the dropped lo*lo term and the split error are ~2^-16 relative per product; accumulation stays fp32).
// v_mfma_f32_32x32x16_bf16 issues 16x the flops per cycle of the fp32 MFMA, so the split runs the contraction 16/3
// faster than exact fp32 at ~fp32 storage traffic.  PREC 0 = exact fp32 MFMA (bitwise an fmaf chain).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// RNE pack of two floats: plain conversions, which the compiler lowers to one v_cvt_pk_bf16_f32 (and, unlike an inline-asm
// form, schedules with the hazard rules of its consumers in mind)
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
// lo = x - float(hi) in one instruction each: v_dot2c_f32_bf16 computes hi.x * (-1) + hi.y * 0 + x (both products and the
// sum are exact: the difference of a float and its bf16 rounding has at most 17 significant bits)
// The multiplier pairs are {-1, -0} and {-0, -1} rather than {-1, 0} / {0, -1}: the latter are folded into the inline
// constant "-1.0", whose placement inside a packed-bf16 operand is not what the fold assumes (measured: wrong results).
// x and y are overwritten with the residuals: the dot2c form accumulates in place, and leaving the inputs dead is what
// lets the compiler do so without a copy per element
__device__ __forceinline__ void split2(float& x, float& y, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(x, y);
  const bf16x2 hv = __builtin_bit_cast(bf16x2, hi);
  x = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, 0x8000BF80u), x, false);
  y = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, 0xBF808000u), y, false);
  lo = pk_bf16(x, y);
}

struct GemmP {
  int M, N, K;
  const float* A; long long lda;
  const float* B; long long ldb;
  float* C; long long ldc;
  int cN, cH, cW, cC, cKH, cKW, cStride, cPad, cOH, cOW, cO;
  const float* scale; const float* bias; const float* residual; long long ldr;
  int relu, accumulate, split_k;
  float* colsum_out;
  uint32_t drop_thresh; float inv_keep;
  const uint64_t* rng_state; uint32_t rng_stream;
  int tilesM, tilesN;
  // the B operand as fragment-ordered bf16 (hi, lo) planes (cape_pack_weights), or null: register-stationary kernel only
  const unsigned short* Bpack;
  // optional gate of the result: v = mask_src[m][n] != 0 ? v * mask_scale : 0 (backward of a fused relu/dropout)
  const float* mask_src; long long ldm; float mask_scale;
  // batched launch (grid.y = batch): batch b = b0 * bdiv + b1 offsets the operands by b0 * s?0 + b1 * s?1 elements
  int bdiv; long long sA0, sA1, sB0, sB1, sC0, sC1;
  int single;                    // bf16x3 kernels only: 1 = one bf16 MFMA per product (hi x hi; precision 2, the measured-only AMP leg)
  int res_cols;                  // 0: residual on every column; else (multiple of 32) on columns < res_cols only
  long long sBias0, sBias1;      // per-batch offsets of `bias`
  // conv-dgrad over a sub-lattice of taps (cape_gemm_desc): column padding, physical filter extent, tap origin / stride
  int cPadX, cKHp, cKWp, cTapH0, cTapHS, cTapW0, cTapWS;
};

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// [k][mn] bf16 image of an mn-contiguous GEMM operand (32 k rows of W = 64 or 128 elements), consumed by ds_read_b64_tr_b16 (gfx950:
// a 16-lane group reads 4 rows x 16 columns and every lane receives one COLUMN -- 4 consecutive k of its mn).  Byte offset of the
// 16-byte chunk `chunk` of row k.  No padding: the chunks of a row are XOR-swizzled by the row so that the 4 rows x 64 bytes a
// 32-lane half reads at once fall on 64 distinct banks (W = 128: rows are 64 banks apart, chunk ^= 4 (k & 3); W = 64: rows are
// 32 banks apart, chunk ^= 4 ((k >> 1) & 1)).
template <int W>
__device__ __forceinline__ int tr_off(int k, int chunk) {
  static_assert(W == 64 || W == 128, "tile widths of the family");
  const int mask = (W == 128) ? ((k & 3) << 2) : (((k >> 1) & 1) << 2);
  return k * (W * 2) + ((chunk ^ mask) << 4);
}
typedef short tr_s4 __attribute__((ext_vector_type(4)));
// 8 consecutive k of the lane's column: two transposed reads 4 rows (`rows4` bytes) apart
__device__ __forceinline__ bf16x8 tr_read8(const char* p, int rows4) {
#if defined(__HIP_DEVICE_COMPILE__)
  const tr_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_s4 __attribute__((address_space(3)))*)(p));
  const tr_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_s4 __attribute__((address_space(3)))*)(p + rows4));
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
#else
  (void)p; (void)rows4;
  return bf16x8{};
#endif
}

constexpr int BK = 32;

// gemm_rs.hip: register-stationary weights (dense A, K in {64, 128, 256}, bf16x3), persistent over 64-row units
bool cape_gemm_rs_eligible(const GemmP& p, int a_mode, int b_mode);
int cape_gemm_rs_launch(const GemmP& p, int b_mode, hipStream_t stream);
