// decode.hip -- device-side token bookkeeping of the autoregressive loop
// (RoomFormerV2.forward_inference, roomformer_v2.py:521-598), so that a decode step needs no
// host round trip and can be captured in a hipGraph.
#include "common.h"

namespace {

__global__ void next_tokens_kernel(const float* cls_logits, const float* reg, int32_t* unfinished, int64_t* tok,
                                   float* delta, const int32_t* step, int N, int nb, int min_len, int eos_id, int sep_id,
                                   int pad_id) {
#pragma clang fp contract(off)   // the reference rounds x*43 before floor/subtract: no FMA contraction here
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int i = step[0];
  // argmax with first-index tie break (torch.argmax)
  const float a = cls_logits[j * 3 + 0], b = cls_logits[j * 3 + 1], c = cls_logits[j * 3 + 2];
  int cls = 0; float best = a;
  if (b > best) { best = b; cls = 1; }
  if (c > best) { best = c; cls = 2; }
  int64_t t11, t12, t21, t22;
  float dx = 0.f, dy = 0.f;
  if (unfinished[j]) {
    if (cls == 0 || (cls == 2 && i < min_len)) {
      const float x = fminf(reg[j * 2 + 0], 1.f), y = fminf(reg[j * 2 + 1], 1.f);
      // explicit rounding steps (no FMA contraction): the reference rounds x*43 before floor and subtract
      const float qx = __fmul_rn(x, (float)(nb - 1)), qy = __fmul_rn(y, (float)(nb - 1));
      const float fx = floorf(qx), fy = floorf(qy), cx = ceilf(qx), cy = ceilf(qy);
      t11 = (int64_t)fx * nb + (int64_t)fy;
      t12 = (int64_t)fx * nb + (int64_t)cy;
      t21 = (int64_t)cx * nb + (int64_t)fy;
      t22 = (int64_t)cx * nb + (int64_t)cy;
      dx = __fsub_rn(qx, fx); dy = __fsub_rn(qy, fy);
    } else if (cls == 1) {
      t11 = t12 = t21 = t22 = sep_id;
    } else {
      unfinished[j] = 0;
      t11 = t12 = t21 = t22 = eos_id;
    }
  } else {
    t11 = t12 = t21 = t22 = pad_id;
  }
  tok[0 * N + j] = t11; tok[1 * N + j] = t12; tok[2 * N + j] = t21; tok[3 * N + j] = t22;
  delta[0 * N + j] = dx; delta[1 * N + j] = __fsub_rn(1.f, dx); delta[2 * N + j] = dy; delta[3 * N + j] = __fsub_rn(1.f, dy);
}

// One block: the token rules above for all N rows (explicit step index, strided logits / coordinates so that a step's slot
// of the output buffers is read in place), the count of still-unfinished rows, and the embedding of the produced tokens for
// the next step (TransformerDecoder._seq_embed, deformable_transformer_v2.py:978-998: four table rows blended by the deltas).
__global__ void __launch_bounds__(256) advance_kernel(const float* cls_logits, long long ld_cls, const float* reg, long long ld_reg,
                                                      int32_t* unfinished, int64_t* tok, float* delta, int step, int N, int nb,
                                                      int min_len, int eos_id, int sep_id, int pad_id, const float* table, int C,
                                                      float* embed_out, int32_t* alive_out) {
#pragma clang fp contract(off)
  __shared__ int alive;
  if (threadIdx.x == 0) alive = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float a = cls_logits[j * ld_cls + 0], b = cls_logits[j * ld_cls + 1], c = cls_logits[j * ld_cls + 2];
    int cls = 0; float best = a;
    if (b > best) { best = b; cls = 1; }
    if (c > best) { best = c; cls = 2; }
    int64_t t11, t12, t21, t22;
    float dx = 0.f, dy = 0.f;
    if (unfinished[j]) {
      if (cls == 0 || (cls == 2 && step < min_len)) {
        const float x = fminf(reg[j * ld_reg + 0], 1.f), y = fminf(reg[j * ld_reg + 1], 1.f);
        const float qx = __fmul_rn(x, (float)(nb - 1)), qy = __fmul_rn(y, (float)(nb - 1));
        const float fx = floorf(qx), fy = floorf(qy), cx = ceilf(qx), cy = ceilf(qy);
        t11 = (int64_t)fx * nb + (int64_t)fy;
        t12 = (int64_t)fx * nb + (int64_t)cy;
        t21 = (int64_t)cx * nb + (int64_t)fy;
        t22 = (int64_t)cx * nb + (int64_t)cy;
        dx = __fsub_rn(qx, fx); dy = __fsub_rn(qy, fy);
      } else if (cls == 1) {
        t11 = t12 = t21 = t22 = sep_id;
      } else {
        unfinished[j] = 0;
        t11 = t12 = t21 = t22 = eos_id;
      }
    } else {
      t11 = t12 = t21 = t22 = pad_id;
    }
    if (unfinished[j]) atomicAdd(&alive, 1);
    tok[0 * N + j] = t11; tok[1 * N + j] = t12; tok[2 * N + j] = t21; tok[3 * N + j] = t22;
    delta[0 * N + j] = dx; delta[1 * N + j] = __fsub_rn(1.f, dx); delta[2 * N + j] = dy; delta[3 * N + j] = __fsub_rn(1.f, dy);
  }
  __syncthreads();                                  // the block's own global writes above are visible to it after the barrier
  if (threadIdx.x == 0 && alive_out) alive_out[0] = alive;
  if (!embed_out) return;
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < N * C4; i += blockDim.x) {
    const int r = i / C4, c = (i - r * C4) * 4;
    // tok rows: [11, 12, 21, 22]; delta rows: [x1, x2, y1, y2]; blend e11*dx2*dy2 + e21*dx1*dy2 + e12*dx2*dy1 + e22*dx1*dy1
    const float dx1 = delta[0 * N + r], dx2 = delta[1 * N + r], dy1 = delta[2 * N + r], dy2 = delta[3 * N + r];
    const float4 e11 = *reinterpret_cast<const float4*>(table + tok[0 * N + r] * C + c);
    const float4 e12 = *reinterpret_cast<const float4*>(table + tok[1 * N + r] * C + c);
    const float4 e21 = *reinterpret_cast<const float4*>(table + tok[2 * N + r] * C + c);
    const float4 e22 = *reinterpret_cast<const float4*>(table + tok[3 * N + r] * C + c);
    float4 o;
    o.x = e11.x * dx2 * dy2 + e21.x * dx1 * dy2 + e12.x * dx2 * dy1 + e22.x * dx1 * dy1;
    o.y = e11.y * dx2 * dy2 + e21.y * dx1 * dy2 + e12.y * dx2 * dy1 + e22.y * dx1 * dy1;
    o.z = e11.z * dx2 * dy2 + e21.z * dx1 * dy2 + e12.z * dx2 * dy1 + e22.z * dx1 * dy1;
    o.w = e11.w * dx2 * dy2 + e21.w * dx1 * dy2 + e12.w * dx2 * dy1 + e22.w * dx1 * dy1;
    *reinterpret_cast<float4*>(embed_out + (long long)r * C + c) = o;
  }
}

}  // namespace

extern "C" int cape_decode_advance(const float* cls_logits, long long ld_cls, const float* reg, long long ld_reg,
                                   int32_t* unfinished, int64_t* tok, float* delta, int step, int N, int num_bins, int min_len,
                                   int eos_id, int sep_id, int pad_id, const float* table, int vocab, int C, float* embed_out,
                                   int32_t* alive_out, cape_stream_t stream) {
  CAPE_REQUIRE(cls_logits && reg && unfinished && tok && delta && step >= 0, "cape_decode_advance: bad arguments");
  CAPE_REQUIRE(ld_cls >= 3 && ld_reg >= 2, "cape_decode_advance: row strides too small");
  if (embed_out) CAPE_REQUIRE(table && C > 0 && C % 4 == 0 && vocab > pad_id && vocab > eos_id && vocab > sep_id &&
                              vocab >= num_bins * num_bins, "cape_decode_advance: embedding table too small for the token ids");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(256), 0, as_stream(stream), cls_logits, ld_cls, reg, ld_reg, unfinished, tok,
                     delta, step, N, num_bins, min_len, eos_id, sep_id, pad_id, table, C, embed_out, alive_out);
  CAPE_LAUNCH_CHECK("cape_decode_advance");
  return 0;
}

extern "C" int cape_decode_next_tokens(const float* cls_logits, const float* reg, int32_t* unfinished, int64_t* tok,
                                       float* delta, const int32_t* step, int N, int num_bins, int min_len, int eos_id,
                                       int sep_id, int pad_id, cape_stream_t stream) {
  CAPE_REQUIRE(cls_logits && reg && unfinished && tok && delta && step, "cape_decode_next_tokens: null pointer");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(next_tokens_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), cls_logits, reg, unfinished, tok,
                     delta, step, N, num_bins, min_len, eos_id, sep_id, pad_id);
  CAPE_LAUNCH_CHECK("cape_decode_next_tokens");
  return 0;
}
