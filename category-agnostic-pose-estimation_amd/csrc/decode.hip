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

}  // namespace

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
