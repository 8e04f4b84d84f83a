/* cape_hip.h -- C ABI of libcape_hip.so, the MI355X (gfx950) kernels of the CAPE episodic
 * training / inference hot path.
 *
 * The reference (nkkrnkl/category-agnostic-pose-estimation) is 100 % PyTorch: it has no FFI or
 * operator-plugin layer.  These entry points are therefore
 * the boundary a maintainer would bind to *replace the torch expressions cited next to each
 * function*; INTEGRATION.md shows the ctypes stub for each.  Conventions:
 *   - plain pointers + sizes only (no torch types); all pointers are DEVICE pointers unless said
 *     otherwise; all tensors fp32 row-major unless said otherwise;
 *   - every function enqueues on `stream` and returns immediately: 0 = ok, nonzero = error
 *     (message via cape_last_error()); no function allocates, frees or synchronises (graph-capturable);
 *   - no ownership transfer; workspaces are explicit arguments;
 *   - thread-compatible: the only global is the thread-local error string.
 */
#ifndef CAPE_HIP_H
#define CAPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cape_stream_t; /* hipStream_t */

const char* cape_last_error(void);
int cape_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * RNG state for dropout: device uint64[2] = {seed, step}.  Masks are a pure function of
 * (seed, step, stream id, element index) so backward regenerates them; cape_rng_advance bumps `step`
 * on device (graph-replay safe).  Replaces torch's global Philox generator used by nn.Dropout
 * (models/deformable_transformer.py:177-183, deformable_transformer_v2.py:286-305).
 * ---------------------------------------------------------------------------------------------- */
int cape_rng_advance(uint64_t* rng_state, cape_stream_t stream);

/* Two-stream ordering for the weight-gradient side stream (runtime of this package; the reference is single-stream):
 * fork = side_stream continues after everything enqueued on main_stream so far, join = the converse.  Graph-capturable. */
int cape_stream_fork(cape_stream_t main_stream, cape_stream_t side_stream);
int cape_stream_join(cape_stream_t main_stream, cape_stream_t side_stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM family on fp32 MFMA (v_mfma_f32_32x32x2_f32):  C[M,N] (+)= epi(A[M,K] * B[K,N])
 *
 * a_mode: 0 dense A[M][K] (lda)                 -- nn.Linear input, 1x1 stride-1 conv input (NHWC)
 *         1 dense A stored transposed [K][M]     -- wgrad: A = dY^T
 *         2 conv-forward im2col gather of an NHWC tensor, k = (kh, kw, c)
 *         3 conv-dgrad gather of dY (NHWC), k = (kh, kw, o)
 * b_mode: 0 dense B stored [N][K] (ldb)          -- nn.Linear / conv weight ([O][KH][KW][C] = channels_last)
 *         1 dense B stored [K][N] (ldb)          -- dgrad of nn.Linear; wgrad's activation operand
 *         2 conv weight read as [(kh,kw,o)][c]   -- conv dgrad
 *         3 conv-wgrad im2col gather, k = output position, n = (kh, kw, c)
 * Epilogue (split_k == 1): v = acc; v = v*scale[n] (opt); v += bias[n] (opt); v += residual[m][n] (opt);
 *   relu (opt); dropout(p) (opt); gate by mask_src (opt, see the struct); then C = v or C += v (accumulate).
 * split_k > 1: partial sums are atomically added into C (C must hold the value to accumulate onto);
 *   the only other epilogue op allowed is the bias (added once, by the first k-split).
 * Replaces: F.linear / nn.Conv2d + FrozenBatchNorm2d (+ReLU, +residual) and their autograd
 *   (models/backbone.py:32-40, torchvision Bottleneck; deformable_transformer.py:95,99-100,113,208;
 *   deformable_transformer_v2.py:314-318,323-331; roomformer_v2.py:192-201,956-968).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int M, N, K;
  int a_mode, b_mode;
  const float* A; long long lda;
  const float* B; long long ldb;
  float* C; long long ldc;
  /* conv geometry for gather modes: input N,H,W,C ; kernel KH,KW,stride,pad ; output OH,OW,O */
  int cN, cH, cW, cC, cKH, cKW, cStride, cPad, cOH, cOW, cO;
  const float* scale;      /* [N] or NULL */
  const float* bias;       /* [N] or NULL */
  const float* residual;   /* [M][ldr] or NULL */
  long long ldr;
  int relu;
  int accumulate;
  int split_k;
  float dropout_p;         /* 0 = off */
  const uint64_t* rng_state;
  uint32_t rng_stream;
  float* colsum_out;       /* a_mode 1 only, or NULL: colsum_out[m] += sum_k A[m][k] -- with A = dY stored [tokens][out] this
                              is the bias gradient of the nn.Linear whose wgrad this GEMM is (atomically accumulated once per
                              k-split; the caller provides the value to accumulate onto) */
  int precision;           /* 0 = exact fp32 MFMA; 1 = bf16x3 split (hi*hi + hi*lo + lo*hi on bf16 MFMA, fp32
                              accumulate, ~2^-16 relative per product); 2 = one bf16 MFMA per product (hi*hi: what
                              torch.autocast(bfloat16) computes -- the measured-only mixed-precision leg of `--use_amp`,
                              engine_cape.py:164-179; misses the 1e-3 logit bar); odd/unaligned shapes always use 0 */
  /* optional (precision 1): the B operand as fragment-ordered bf16 (hi, lo) planes written by cape_pack_weights for this
     (N, K, ldb, b_mode).  Only the register-stationary kernel (dense A, K in {64, 128, 256}) reads it: its per-block weight
     prologue becomes K/8 coalesced loads per wave instead of a load -> split -> LDS -> fragment pass; every other shape
     ignores the field.  Must describe the same matrix as B / b_mode (B itself stays required). */
  const uint16_t* B_packed;
  /* optional gate applied last (split_k == 1): v = mask_src[m][n] != 0 ? v * mask_scale : 0, mask_src (M, ldm).  With
     mask_src = the saved output of a fused linear+ReLU(+dropout p) and mask_scale = 1/(1-p), the dgrad of the *next*
     layer emits the pre-activation gradient directly (FFN backward without a separate relu/dropout-backward pass). */
  const float* mask_src; long long ldm; float mask_scale;
  /* optional batched launch (dense modes, no epilogue vectors, split_k 1): `batch` products in one launch, batch index
     b = b0 * batch_div + b1 offsets A / B / C by b0 * s?0 + b1 * s?1 elements -- e.g. the per-(image, head) Q K^T and P V
     products of attention with heads interleaved in the rows (b0 = image, b1 = head, s?1 = 32). batch <= 1 = single product. */
  int batch, batch_div; long long sA0, sA1, sB0, sB1, sC0, sC1;
  /* round 3 (ABI 8).  res_cols: 0 = the residual applies to every column; else (a multiple of 32) only to columns < res_cols
     -- q | k | v of the decoder's self-attention as ONE product over the stacked attn_q / attn_k / attn_v rows, with
     `+ query_pos` on the q columns only (deformable_transformer_v2.py:323-331).  sBias0 / sBias1: per-batch offsets of `bias`
     for batched launches (the three in_proj blocks of nn.MultiheadAttention as one batch-3 launch). */
  int res_cols; long long sBias0, sBias1;
  /* round 3 (ABI 11), conv-dgrad (a_mode 3 / b_mode 2, stride 1, O % 32 == 0) over a SUB-LATTICE of filter taps.  A stride-2
     convolution's data gradient falls into four input-parity classes; each is a stride-1 data gradient on the half-resolution
     grid whose taps are a strided subset of the physical filter (3x3, pad 1: {0,2} for odd rows / columns, {1} for even ones):
     cKH / cKW = taps of the class, cPad / cPadX = its row / column padding, and tap (kh', kw') reads the physical filter tap
     (cTapH0 + kh' * cTapHS, cTapW0 + kw' * cTapWS) of a (cKHp x cKWp) filter.  cKHp == 0: the filter is cKH x cKW as given
     and cPadX = cPad (every earlier caller).  The whole gradient does 2.25 / 9 of the multiplications of the one-launch form. */
  int cPadX, cKHp, cKWp, cTapH0, cTapHS, cTapW0, cTapWS;
} cape_gemm_desc;

int cape_gemm_f32(const cape_gemm_desc* d, cape_stream_t stream);

/* Grouped launch: `n` (1..CAPE_GEMM_GROUP_MAX) independent products of the family in ONE launch; `descs` is a HOST array (the
 * item table travels in the kernel arguments: nothing to keep alive, capturable).  For the weight gradients of a training
 * step -- dW = dY^T X (torch autograd of F.linear / nn.Conv2d: deformable_transformer.py:95-113,208-231, backbone.py:32-40) --
 * which nothing waits for until the optimizer step, so the host queues them during the backward pass and submits them 8-32
 * at a time: the tiles of all items fill the chip together and each item needs few k-splits (few float atomics).
 * Every item: a_mode 1, the same b_mode (1 dense [K][N], or 3 conv-wgrad im2col), the same precision, accumulate = 1
 * (split_k > 1: atomic partial sums; split_k == 1: C += tile), optional colsum_out, no other epilogue operand, no batch;
 * 16-byte aligned operands with M, N, lda, ldb multiples of 4.  `tile` = 64 or 128 (output tile edge of every item). */
#define CAPE_GEMM_GROUP_MAX 32
int cape_gemm_group_f32(const cape_gemm_desc* descs, int n, int tile, cape_stream_t stream);

/* Weight packing for cape_gemm_desc.B_packed.  An item describes one weight as a B operand (b_mode 0: stored [N][K], nn.Linear
 * forward; b_mode 1: stored [K][N], its dgrad) and the destination of cape_packed_weight_bytes(N, K) bytes.  `items_dev` is a
 * DEVICE array, so the table of all weights of a model is uploaded once and one launch per optimizer step re-packs them. */
typedef struct { const float* B; uint16_t* out; long long ldb; int N, K, b_mode, pad; } cape_pack_item;
size_t cape_packed_weight_bytes(int N, int K);
int cape_pack_weights(const cape_pack_item* items_dev, int n_items, int max_blocks_per_item, cape_stream_t stream);

/* column sums over nbatch row blocks:  out[n] (+)= sum_b sum_m X[b*batch_stride + m*ldx + n]
 * (bias gradients: nbatch = 1; level_embed gradient: one block of a level's rows per image) */
int cape_colsum_f32(const float* X, long long ldx, int nbatch, long long batch_stride, int M, int N, float* out,
                    int accumulate, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Cached autoregressive decode step (inference): N <= 64 token rows against the decoder's weights, exact fp32 FMA.
 *
 * cape_decode_linear:  out = [relu]( LNin(X) [+ in_add] ) W^T [+ X2 W2^T on the first n2 columns] + bias [+ LNres(R)]
 *   X (N, K) rows; in_gamma/in_beta != NULL: X holds a *pre-norm* sum and is layer-normalised (eps 1e-5) while it is staged
 *   ("LayerNorm on load": the post-norm chain x' = LN(x + f(x)) of deformable_transformer_v2.py:340-365 is kept as
 *   pre-norm sums, each consumer normalises the rows it reads); in_add (N, K) is added after the norm (`+ query_pos`);
 *   W (Nout, K) as stored by nn.Linear; X2 (N, K2) / W2 (n2, K2): a second product for columns < n2 (n2 % 8 == 0);
 *   R (N, Nout): residual, layer-normalised with res_gamma/res_beta when given;
 *   the Nout columns are written as nseg (<= 3) segments of `seg` columns, segment s to out[s] with row stride ldo[s]
 *   (q | k | v of a layer in one launch, k and v straight into row `step` of the KV cache: kv_cache.py:21-36).
 * Replaces per layer: attn_q/k/v + MultiheadAttention in_proj (folded, deformable_transformer_v2.py:323-331), out_proj,
 *   support_attn in/out projections, sampling_offsets|attention_weights, output_proj, linear1, linear2 and norm2 /
 *   norm_support / norm1 / norm3 (deformable_transformer_v2.py:320-370; deformable_transformer.py:99-113).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int N, K, Nout;
  const float* X; long long ldx; const float* in_gamma; const float* in_beta; const float* in_add; long long ld_add;
  const float* W; long long ldw; const float* bias;
  const float* X2; long long ldx2; int K2; const float* W2; long long ldw2; int n2;
  const float* R; long long ldr; const float* res_gamma; const float* res_beta;
  int relu;
  int nseg, seg; float* out[3]; long long ldo[3];
} cape_decode_linear_desc;
int cape_decode_linear(const cape_decode_linear_desc* d, cape_stream_t stream);

/* cape_decode_tail: what lies between two decoder layers for one token per image, one launch:
 *   t = LN3(P4) (-> hs_out when given);  delta = MLP(t) (W1, W2: 256x256 + ReLU, W3: 2x256);
 *   ref' = sigmoid(delta + inverse_sigmoid(ref)) (eps 1e-5 clamps, util/misc.py:436-440) -> ref_out (row stride ld_ref);
 *   Wc != NULL (last layer): class logits Wc t + Bc -> cls_out (row stride ld_cls);
 *   Wp != NULL (a next layer exists): qpos_out = LN(pos_trans(sine256(ref' * 2 pi)); gp, bp) and
 *     refin_out[n][l] = ref' * valid_ratio[n][l]  (deformable_transformer_v2.py:1077-1090, :1000-1018).
 * Replaces TransformerDecoder.forward :1083-1110 (per layer) + MLP (roomformer_v2.py:956-968) for Lq = 1. */
typedef struct {
  int N, L;
  const float* P4; long long ldp; const float* g3; const float* b3;
  const float* W1; const float* B1; const float* W2; const float* B2; const float* W3; const float* B3;
  const float* ref;
  const float* Wc; const float* Bc; int ncls;
  const float* Wp; const float* Bp; const float* gp; const float* bp;
  const float* dim_t; const float* vr;
  float* ref_out; long long ld_ref;
  float* qpos_out; float* refin_out;
  float* cls_out; long long ld_cls;
  float* hs_out; long long ld_hs;
} cape_decode_tail_desc;
int cape_decode_tail(const cape_decode_tail_desc* d, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * cape_decode_step: ONE launch for a whole cached decode step -- every decoder layer, for one query token per image
 * (one 512-thread block per image; nothing in a step couples two images, so the chain of ~20 dependent matrix-vector
 * stages per layer needs no grid-level synchronisation; weights stream through L2 into each block with the next stage's
 * block always requested ahead).  Replaces the per-stage launches of cape_decode_linear / cape_attn_fwd / cape_msda_fwd /
 * cape_decode_tail for one step of TransformerDecoder.forward (deformable_transformer_v2.py:1024-1131) with
 * TransformerDecoderLayer v1 (:320-370), as driven by RoomFormerV2.forward_inference (roomformer_v2.py:481-598):
 *   per layer l:  q|k|v = x W_qkv^T + b (folded attn_{q,k,v} . in_proj), q += query_pos W_qin^T; k, v -> row `step` of the
 *   layer's cache; single-query self-attention over rows 0..step (8 heads x 32); out_proj + residual, norm2; [support
 *   cross-attention over P cached keys with key-padding mask, out_proj + residual, norm_support]; sampling_offsets |
 *   attention_weights of (t + query_pos); softmax over L*n_points = 16 logits per head + bilinear gather from the cached
 *   value projection (N, S, 256); output_proj + residual, norm1; linear1 + ReLU, linear2 + residual; norm3; coords MLP
 *   (256-256-256-2) + refinement sigmoid(delta + logit(ref)); [last layer: class head, hidden state] / [else: next layer's
 *   query position embedding LN(pos_trans(sine(ref'))) and level-scaled reference points ref' * valid_ratio].
 * All matrices row-major [out][in] fp32, 16-byte aligned.  Model width 256, 8 heads, L * n_points == 16, ffn_dim a multiple
 * of 256 (<= 1024), S < 65535, cache rows T <= 1024, P <= 1024.  w_sq == NULL: no support attention in that layer.
 * ---------------------------------------------------------------------------------------------- */
#define CAPE_DECODE_MAX_LAYERS 8
typedef struct {
  const float *w_qkv, *b_qkv;            /* (768, 256) folded projection, (768,) in_proj_bias */
  const float* w_qin;                    /* (256, 256) in_proj_weight[:256]: carries `+ query_pos` of the query */
  float *k_cache, *v_cache;              /* (N, T, 256) */
  const float *w_o, *b_o, *ln2_g, *ln2_b;
  const float *w_sq, *b_sq;              /* support attention: query projection (in_proj rows 0..255) or NULL */
  const float *sup_k, *sup_v;            /* (N, P, 256) projected support keys / values */
  const unsigned char* sup_mask;         /* (N, P) nonzero = padded key, or NULL */
  const float *w_so, *b_so, *lns_g, *lns_b;
  const float *w_off, *b_off;            /* (384, 256) sampling_offsets | attention_weights, (384,) */
  const float* value;                    /* (N, S, 256) cached value projection of the image memory */
  const float *w_mo, *b_mo, *ln1_g, *ln1_b;
  const float *w1, *b1, *w2, *b2;        /* (F, 256), (F,), (256, F), (256,) */
  const float *ln3_g, *ln3_b;
  const float *m1w, *m1b, *m2w, *m2b, *m3w, *m3b;   /* coords MLP of this layer: (256,256) (256,256) (2,256) */
} cape_decode_layer_desc;
typedef struct {
  int N, n_layers, step, T, P, S, L, n_points, ncls, ffn_dim;
  int shapes[8];                         /* (H_l, W_l) per level */
  int level_start[4];
  const float* emb;                      /* (N, 256) embedding of the step's input tokens */
  const float* qpos0;                    /* (256,) layer-0 query position embedding of this step (same for all images) */
  const float* refin0;                   /* (N, L, 2) layer-0 level-scaled reference points */
  const float* ref0;                     /* (N, 2) layer-0 reference points */
  const float* vr;                       /* (N, L, 2) valid ratios */
  const float* dim_t;                    /* (128,) sine periods */
  const float *class_w, *class_b;        /* (ncls, 256), (ncls,): class head of the last layer */
  const float *pos_w, *pos_b, *pos_gamma, *pos_beta;   /* pos_trans + pos_trans_norm (shared by the layers) */
  float* out_logits; long long ld_logits;   /* row n of this step's slot: out_logits + n * ld_logits, ncls values */
  float* out_coords; long long ld_coords;   /* 2 values */
  float* out_hs; long long ld_hs;           /* 256 values: last layer's normalised hidden state */
  cape_decode_layer_desc layers[CAPE_DECODE_MAX_LAYERS];
} cape_decode_step_desc;
int cape_decode_step(const cape_decode_step_desc* d, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * out = LayerNorm(x + dropout(y)) * gamma + beta over rows of C (C <= 1024, C % 4 == 0), eps 1e-5.
 * y may be NULL (plain LayerNorm).  Saves mean/rstd (per row) for backward.  If pos != NULL also writes
 * out_pos = out + pos (the `with_pos_embed` add, deformable_transformer.py:197).
 * Replaces `self.normK(src + self.dropoutK(src2))` (deformable_transformer.py:209-210,228-229;
 * deformable_transformer_v2.py:340-341,356-357,364-365,316-317) and nn.TransformerEncoderLayer's norms.
 * ---------------------------------------------------------------------------------------------- */
int cape_add_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta,
                           float* out, float* mean, float* rstd, const float* pos, float* out_pos,
                           int rows, int C, float dropout_p, const uint64_t* rng_state, uint32_t rng_stream,
                           cape_stream_t stream);
/* backward: given d_out (and optional d_out_pos, added), recomputes s = x + dropout(y) and returns
 * d_x (= ds) and d_y (= ds * mask / keep); accumulates dgamma/dbeta (+=).  d_y may alias d_x when y == NULL
 * or dropout_p == 0 (then only d_x is written and d_y must be NULL or equal to d_x). */
int cape_add_layernorm_bwd(const float* d_out, const float* d_out_pos, const float* x, const float* y,
                           const float* gamma, const float* mean, const float* rstd,
                           float* d_x, float* d_y, float* dgamma, float* dbeta,
                           int rows, int C, float dropout_p, const uint64_t* rng_state, uint32_t rng_stream,
                           cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(G groups) over NHWC data x[n][hw][C] -> out written with a per-image row stride so that
 * the result lands directly inside the flattened multi-level token buffer (N, S, C) at the level's
 * offset.  eps 1e-5.  Replaces nn.GroupNorm(32, hidden_dim) + flatten(2).transpose(1,2) + cat
 * (roomformer_v2.py:192-201, deformable_transformer_v2.py:189-200).
 * ---------------------------------------------------------------------------------------------- */
/* workspace: cape_groupnorm_workspace_bytes(N, C, G) bytes of device memory (8-byte aligned, contents scratch: the
 * per-(image, group) fp64 sums of the forward, the per-(image, channel) sums of the backward). */
size_t cape_groupnorm_workspace_bytes(int N, int C, int G);
int cape_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* out,
                       long long out_image_stride, float* mean, float* rstd,
                       int N, int HW, int C, int G, void* workspace, size_t workspace_bytes, cape_stream_t stream);
int cape_groupnorm_bwd(const float* d_out, long long d_out_image_stride, const float* x,
                       const float* gamma, const float* mean, const float* rstd,
                       float* d_x, float* dgamma, float* dbeta, int N, int HW, int C, int G,
                       void* workspace, size_t workspace_bytes, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-scale deformable attention core (fused softmax over L*P logits + bilinear gather + weighted sum).
 *   value   (N, S, M, D)        D == 32, M == 8
 *   offw    (N, Lq, 384)  = [ sampling_offsets (M,L,P,2) | attention logits (M,L*P) ] per query
 *   ref     (N, Lq, L, 2)       normalised reference points
 *   shapes  int32 [L][2] (H_l, W_l) ; level_start int32 [L]        (host pointers)
 *   out     (N, Lq, M*D)
 * Semantics of F.grid_sample(bilinear, zeros padding, align_corners=False) on grid 2*loc-1.
 * Replaces MSDeformAttn.forward:99-112 + ms_deform_attn_core_pytorch (deformable_transformer.py:115-141).
 * ---------------------------------------------------------------------------------------------- */
int cape_msda_fwd(const float* value, const float* offw, const float* ref, const int* shapes,
                  const int* level_start, float* out, int N, int S, int Lq, int L, int P,
                  cape_stream_t stream);
/* backward: d_value, d_offw and d_ref (may be NULL) are fully written (no caller-side zero fill needed). */
int cape_msda_bwd(const float* d_out, const float* value, const float* offw, const float* ref,
                  const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                  int N, int S, int Lq, int L, int P, cape_stream_t stream);
/* the same backward with the accumulator type of the d_value slab chosen by the caller: value_accum 0 = fp64 LDS accumulators
 * (8 channels per block; what cape_msda_bwd does: the sum is rounded to fp32 once), 1 = 64-bit integer accumulators holding two
 * channels each as block-scaled 32-bit fixed point (16 channels per block, ~2x faster; quantum <= pow2ceil(Lq) * 2^-29 of the
 * block's largest |d_out|, cannot overflow; a non-finite d_out makes the block's d_value slice NaN).  The product path uses 1
 * with the bf16x3 / bf16 GEMM modes and 0 on the exact-fp32 leg. */
int cape_msda_bwd_ex(const float* d_out, const float* value, const float* offw, const float* ref,
                     const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                     int N, int S, int Lq, int L, int P, int value_accum, cape_stream_t stream);
/* any-geometry form of the same backward: memory-side float atomics into d_value (what cape_msda_bwd falls back to when
 * the (S+1) x 8 fp64 gradient slab of one (image, head, channel group) does not fit in LDS). */
int cape_msda_bwd_atomic(const float* d_out, const float* value, const float* offw, const float* ref,
                         const int* shapes, const int* level_start, float* d_value, float* d_offw, float* d_ref,
                         int N, int S, int Lq, int L, int P, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small dense attention core, heads of 32 channels packed in rows of `ld` floats:
 *   O[n,i,h,:] = sum_j softmax_j(scale * Q[n,i,h,:].K[n,j,h,:] + mask) (dropped) V[n,j,h,:]
 * mask_mode 0 none, 1 causal (j <= i + causal_offset), 2 key padding (uint8 kpm[n][Lk], 1 = ignore).
 * lse (N,H,Lq) saved for backward.  Fully masked rows give NaN exactly like torch softmax(-inf row).
 * Replaces the core of nn.MultiheadAttention (deformable_transformer_v2.py:339, :352-355;
 * geometric_support_encoder.py:223-226).
 * ---------------------------------------------------------------------------------------------- */
int cape_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse,
                  long long ldq, long long ldk, long long ldv, long long ldo,
                  long long bsq, long long bsk, long long bsv, long long bso, /* batch strides (elements) */
                  int N, int H, int Lq, int Lk, float scale, int mask_mode, int causal_offset,
                  const uint8_t* kpm, float dropout_p, const uint64_t* rng_state, uint32_t rng_stream,
                  cape_stream_t stream);
int cape_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* O,
                  const float* lse, float* dQ, float* dK, float* dV,
                  long long ldq, long long ldk, long long ldv, long long ldo,
                  long long bsq, long long bsk, long long bsv, long long bso,
                  int N, int H, int Lq, int Lk, float scale, int mask_mode, int causal_offset,
                  const uint8_t* kpm, float dropout_p, const uint64_t* rng_state, uint32_t rng_stream,
                  cape_stream_t stream);
/* Row softmax of the matrix-core attention form: the contractions Q K^T / P V (and dV, dP, dQ, dK) are batched
 * cape_gemm_f32 launches, one product per (image, head); S, P, Pd, dS are (N, H, Lq, Lk) fp32, masks and the dropout
 * stream indexed exactly as in cape_attn_fwd/bwd.  fwd: P = softmax(scale*S + mask), Pd = dropout(P) (Pd NULL iff p = 0);
 * bwd (in place on dS, which holds dPd on entry): dS = scale * P o (g - sum_j g P), g = dropout-mask o dPd / keep. */
int cape_attn_softmax_fwd(const float* S, float* P, float* Pd, int N, int H, int Lq, int Lk, float scale, int mask_mode,
                          int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
                          uint32_t rng_stream, cape_stream_t stream);
int cape_attn_softmax_bwd(const float* P, float* dS, int N, int H, int Lq, int Lk, float scale, float dropout_p,
                          const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream);

/* Fused attention core on the matrix cores (csrc/flash_attn.hip; round 3): rows of up to 224 keys, head dim 32, bf16x3 split.
 * One launch computes O = dropout(softmax(scale Q K^T + mask)) V and the log-sum-exp `lse` (N, H, Lq) of the scaled, masked
 * scores -- the (N, H, Lq, Lk) score / probability tensors never leave registers.  cape_flash_attn_bwd recomputes P from
 * `lse` (two launches: dQ and D = rowsum(dO o O) into `d_ws` (N*H*Lq floats); then dK | dV).  Layouts, strides, mask modes
 * (0 none, 1 causal with offset, 2 key padding) and the dropout element index are those of cape_attn_fwd / cape_attn_bwd;
 * dQ / dK / dV use the strides of Q / K / V, dO those of O.  All row strides multiples of 4 floats, bases 16-byte aligned.
 * Replaces the core of nn.MultiheadAttention in the decoder's causal self-attention (deformable_transformer_v2.py:323-341;
 * torch: F.scaled_dot_product_attention semantics with attention dropout). */
int cape_flash_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, long long ldq, long long ldk,
                        long long ldv, long long ldo, long long bsq, long long bsk, long long bsv, long long bso, int N, int H,
                        int Lq, int Lk, float scale, int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p,
                        const uint64_t* rng_state, uint32_t rng_stream, cape_stream_t stream);
int cape_flash_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* O, const float* lse,
                        float* dQ, float* dK, float* dV, float* d_ws, long long ldq, long long ldk, long long ldv, long long ldo,
                        long long bsq, long long bsk, long long bsv, long long bso, int N, int H, int Lq, int Lk, float scale,
                        int mask_mode, int causal_offset, const uint8_t* kpm, float dropout_p, const uint64_t* rng_state,
                        uint32_t rng_stream, cape_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Elementwise / small ops.  `dim_t` = device float[128] temperature table
 *   10000 ** (2*(k//2)/128) computed by the host with the same torch expression as the reference.
 * ---------------------------------------------------------------------------------------------- */
/* out = a + b (n floats, n % 4 == 0 not required) */
int cape_add_f32(const float* a, const float* b, float* out, long long n, cape_stream_t stream);
/* out[n][s][:] = base[n][s][:] + level_embed[level of token s][:]  (base, out (N, S, C); level_start[l] = first token of level l).
 * The sine part of the image position embedding is a constant of the geometry for unpadded batches (cached by the host side); this
 * adds the trainable `level_embed` row of each level (deformable_transformer_v2.py:196 `lvl_pos_embed = pos_embed + level_embed[lvl]`). */
int cape_level_embed_add(const float* base, const float* level_embed, const int* level_start, float* out, int N, int S, int L, int C,
                         cape_stream_t stream);
/* out = gelu(x), exact erf form (nn.GELU default; timm Mlp act of models/bixattn.py:116-125) */
/* out = srcs[0] + ... + srcs[k-1] (k <= 8 host-array of device pointers, n elements each): the gradient fan-in of a tensor with
 * several consumers in one pass (what autograd's InputBuffer does with k-1 `at::add` launches). */
int cape_add_n_f32(const float* const* srcs, int k, float* out, long long n, cape_stream_t stream);
/* the same for (rows, cols) sources with row strides lds[j] (cols % 4 == 0): a summand may be a column block of a wider buffer,
 * and so may the output (row stride ldo; `out` may coincide with a source: in-place accumulation) */
/* out[n][y][x][:] = (acc ? acc[n][y][x][:] : 0) + cls[(y & 1) * 2 + (x & 1)][n][y >> 1][x >> 1][:] for NHWC tensors (H, W even): the
 * four input-parity classes of a stride-2 convolution's data gradient (cape_gemm_desc, sub-lattice fields) back on the full grid;
 * a NULL class contributes zeros (1x1 stride-2: only the (even, even) class exists); `acc` may be `out` (in place). */
int cape_interleave2x2_f32(const float* const* cls, const float* acc, float* out, int N, int H, int W, int C, cape_stream_t stream);
int cape_add_n_rows_f32(const float* const* srcs, const long long* lds, int k, float* out, long long ldo, long long rows, int cols,
                        cape_stream_t stream);
int cape_gelu_f32(const float* x, float* out, long long n, cape_stream_t stream);
/* out[r][c] = x[r][c] + y[r][c] * gamma[c] (gamma may be NULL): residual behind LayerScale (models/bixattn.py:5-31,135-141) */
int cape_scale_residual_f32(const float* x, const float* y, const float* gamma, float* out, long long rows, int C,
                            cape_stream_t stream);
/* backward of the exact GELU (bixattn.py Mlp): dx = g * (Phi(x) + x phi(x)) */
int cape_gelu_bwd_f32(const float* x, const float* g, float* dx, long long n, cape_stream_t stream);
/* backward of out = x + gamma * y (LayerScale residual, bixattn.py:11-19) w.r.t. y and gamma: dy = gamma * g,
 * dgamma[c] += sum_r g[r][c] y[r][c] (dgamma is accumulated into: zero or hold a running gradient) */
int cape_scale_residual_bwd_f32(const float* g, const float* y, const float* gamma, float* dy, float* dgamma, long long rows, int C,
                                cape_stream_t stream);
/* NCHW (N,C,H,W) -> NHWC with channel padding to Cp (zeros) */
int cape_nchw_to_nhwc(const float* x, float* out, int N, int C, int H, int W, int Cp, cape_stream_t stream);
/* FrozenBatchNorm fold: scale = w * rsqrt(rv + eps), shift = b - rm * scale  (backbone.py:32-40) */
int cape_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps,
                 float* scale, float* shift, int C, cape_stream_t stream);
/* 3x3 stride-2 pad-1 max pool on NHWC */
int cape_maxpool3x3s2_nhwc(const float* x, float* out, int N, int H, int W, int C, cape_stream_t stream);
/* backward through y = relu(conv*scale + shift (+res)):  d_pre = dy * (y > 0) * scale[c] ;
 * d_res = dy * (y > 0) (optional).  relu == 0 skips the mask.  rows x C, C % 4 == 0. */
int cape_bn_relu_bwd(const float* dy, const float* y, const float* scale, float* d_pre, float* d_res,
                     long long rows, int C, int relu, cape_stream_t stream);
/* in place: x = [relu](x * scale[c] + bias[c] [+ residual]) over (rows, C): FrozenBatchNorm2d affine + ReLU + shortcut
 * (backbone.py:32-40, torchvision Bottleneck) as a separate pass, for convolutions whose contraction was split over k. */
int cape_affine_act_f32(float* x, const float* scale, const float* bias, const float* residual, long long rows, int C, int relu,
                        cape_stream_t stream);
/* d_pre = dh * (h > 0) * inv_keep  (backward of relu+dropout fused in a GEMM epilogue) */
int cape_relu_drop_bwd(const float* dh, const float* h, float* d_pre, long long n, float inv_keep,
                       cape_stream_t stream);
/* image sine position embedding + level embedding, written into the flattened (N,S,256) buffer:
 * models/position_encoding.py:22-40 with normalize=True; mask uint8 (N,h,w) 1 = padded. */
int cape_pos_sine_level(const uint8_t* mask, const float* level_embed_l, const float* dim_t, float* out,
                        long long out_image_stride, int N, int h, int w, int C, cape_stream_t stream);

/* 4-corner bilinear token embedding (deformable_transformer_v2.py:984-997):
 * ids int64 (R) x4, deltas float (R) x4 -> out (R, C) */
int cape_token_embed_fwd(const float* table, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                         const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                         const float* dy2, float* out, long long R, int C, int vocab, cape_stream_t stream);
int cape_token_embed_bwd(const float* d_out, const int64_t* s11, const int64_t* s21, const int64_t* s12,
                         const int64_t* s22, const float* dx1, const float* dx2, const float* dy1,
                         const float* dy2, float* d_table, long long R, int C, int vocab, int pad_idx,
                         cape_stream_t stream);

/* decoder query sine embedding (deformable_transformer_v2.py:1005-1018): ref (R,2) -> out (R,256),
 * x-block then y-block, 128 features per axis */
int cape_query_sine_fwd(const float* ref, const float* dim_t, float* out, long long R, cape_stream_t stream);
int cape_query_sine_bwd(const float* d_out, const float* ref, const float* dim_t, float* d_ref, int accumulate, long long R,
                        cape_stream_t stream);

/* iterative refinement (deformable_transformer_v2.py:1096-1102, util/misc.py:436-440):
 * new_ref = sigmoid(delta + inverse_sigmoid(ref)), eps 1e-5 ; n elements */
int cape_refine_fwd(const float* delta, const float* ref, float* new_ref, long long n, cape_stream_t stream);
/* d_delta = d_new * s(1-s) ; d_ref (+)= d_new * s(1-s) * d inverse_sigmoid(ref)/d ref */
int cape_refine_bwd(const float* d_new, const float* new_ref, const float* ref, float* d_delta,
                    float* d_ref, int accumulate_ref, long long n, cape_stream_t stream);
/* y = sigmoid(x) and backward dx (+)= dy * y (1-y) */
int cape_sigmoid_fwd(const float* x, float* y, long long n, cape_stream_t stream);
int cape_sigmoid_bwd(const float* dy, const float* y, float* dx, int accumulate, long long n,
                     cape_stream_t stream);
/* ref_in[r][l][:] = ref[r][:] * valid_ratio[n(r)][l][:]  (R rows, rows_per_image rows per image) and bwd */
int cape_ref_scale_fwd(const float* ref, const float* valid_ratios, float* ref_in, long long R,
                       int rows_per_image, int L, cape_stream_t stream);
int cape_ref_scale_bwd(const float* d_ref_in, const float* valid_ratios, float* d_ref, int accumulate,
                       long long R, int rows_per_image, int L, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Geometric support encoder pieces (models/geometric_support_encoder.py:163-192, graph_utils.py)
 * ---------------------------------------------------------------------------------------------- */
/* h = relu(coords @ W0^T + b0) (R,256) ; pe = sine2d(coords) + pe1d[p]  (R,256) ; R = N*P rows */
int cape_support_embed_fwd(const float* coords, const float* W0, const float* b0, const float* pe1d,
                           const float* dim_t, float* h, float* pe, int N, int P, int C, cape_stream_t stream);
/* dW0 (+)= dh^T coords masked by relu ; db0 (+)= ... ; (coords carry no gradient) */
int cape_support_embed_bwd(const float* d_h, const float* h, const float* coords, float* dW0, float* db0,
                           int N, int P, int C, cape_stream_t stream);
/* adjacency (N,2,P,P) from an edge list: edges int32 (E_total,2), edge_start int32 (N+1), mask uint8 (N,P)
 * (1 = ignore).  graph_utils.py:46-80 */
int cape_adjacency(const int* edges, const int* edge_start, const uint8_t* mask, float* adj, int N, int P,
                   cape_stream_t stream);
/* GCN aggregate: out[n,w,c] = relu( sum_k sum_v adj[n,k,v,w] * y[n,v,k*C+c] )  (graph_utils.py:157-186) */
/* key-padding glue of GeometricSupportEncoder.forward (geometric_support_encoder.py:201-220) in one launch: mask (N, P) u8,
 * non-zero = ignore -> kpm = mask with keypoint 0 unmasked where a graph is fully masked, zero = rows to be zeroed afterwards
 * (fully masked graphs; with pad_rows also every masked row: the nested-tensor fast path of nn.TransformerEncoder). */
int cape_support_masks(const uint8_t* mask, uint8_t* kpm, uint8_t* zero, int N, int P, int pad_rows, cape_stream_t stream);
int cape_gcn_aggregate_fwd(const float* y, const float* adj, float* out, int N, int P, int C,
                           cape_stream_t stream);
/* d_y[n,v,k*C+c] = sum_w adj[n,k,v,w] * d_out[n,w,c] * (out[n,w,c] > 0) */
int cape_gcn_aggregate_bwd(const float* d_out, const float* out, const float* adj, float* d_y, int N,
                           int P, int C, cape_stream_t stream);
/* rows with rowmask[r] != 0 are set to zero (n floats per row) */
int cape_zero_rows(float* x, const uint8_t* rowmask, long long rows, int C, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Loss (models/cape_losses.py:71-163, roomformer_v2.py:915-953), all NL decoder layers at once.
 *   logits (NL, R, 3), coords (NL, R, 2), labels int64 (R), vis uint8 (R), target (R, 2)
 *   class_w float[3]; w_ce, w_l1 loss weights; loss_scale multiplies the gradients (1/accum steps)
 * Outputs: losses float[2*NL] = {ce_l, l1_l}_l (unweighted), total float[1] (weighted sum),
 *          d_logits / d_coords = d total*loss_scale / d input.
 * ---------------------------------------------------------------------------------------------- */
int cape_loss_fwd_bwd(const float* logits, const float* coords, const int64_t* labels, const uint8_t* vis,
                      const float* target, const float* class_w, float w_ce, float w_l1, float loss_scale,
                      float* losses, float* total, float* d_logits, float* d_coords, int NL, long long R,
                      cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer over flat arenas (train_cape_episodic.py:527-538, engine_cape.py:240-258):
 *   cape_sumsq: out[b] = partial sum of g^2 of block b, b < CAPE_SUMSQ_PARTS (no atomics: the reduction order is fixed, so data-
 *   parallel replicas that hold the same all-reduced gradient compute the same clip coefficient bit for bit);
 *   cape_adamw_step: clip coefficient min(1, max_norm / (sqrt(sum of the n_parts partial sums) + 1e-6)) read from the device,
 *   torch.optim.AdamW semantics (decoupled weight decay, bias correction with `step` read from device step_count[0],
 *   incremented by cape_step_increment).
 * ---------------------------------------------------------------------------------------------- */
#define CAPE_SUMSQ_PARTS 256
int cape_sumsq(const float* g, long long n, float* out_parts, cape_stream_t stream);
int cape_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float max_norm, const float* sumsq_parts, int n_parts,
                    const int64_t* step_count, const float* lr_dev /* device scalar overriding `lr` when not NULL: a captured
                    step follows the learning-rate schedule (torch.optim.lr_scheduler writes param_groups[i]["lr"]) */,
                    cape_stream_t stream);
int cape_step_increment(int64_t* step_count, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Autoregressive decode bookkeeping on device (roomformer_v2.py:521-598): from the step's class logits
 * (N,3) and coordinates (N,2) produce the next step's 4 token ids + 4 deltas and update the unfinished
 * flags; also appends logits/coords to the per-step output buffers.
 * ---------------------------------------------------------------------------------------------- */
/* The same rules with the step index as an argument, strided logits / coordinates (a slot of the (N, T, 3) / (N, T, 2) output
 * buffers), alive_out[0] = rows still unfinished afterwards, and -- when embed_out != NULL -- the embedding of the produced
 * tokens for the next step (TransformerDecoder._seq_embed, deformable_transformer_v2.py:978-998; table (vocab, C)). */
int cape_decode_advance(const float* cls_logits, long long ld_cls, const float* reg, long long ld_reg, int32_t* unfinished,
                        int64_t* tok, float* delta, int step, int N, int num_bins, int min_len, int eos_id, int sep_id, int pad_id,
                        const float* table, int vocab, int C, float* embed_out, int32_t* alive_out, cape_stream_t stream);
int cape_decode_next_tokens(const float* cls_logits, const float* reg, int32_t* unfinished,
                            int64_t* tok /* (4,N): 11,12,21,22 */, float* delta /* (4,N): x1,x2,y1,y2 */,
                            const int32_t* step /* device scalar */, int N, int num_bins, int min_len,
                            int eos_id, int sep_id, int pad_id, cape_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GPU side of the MP-100 loader (csrc/augment.hip): raw uint8 crops -> augmented, resized, normalised
 * (3, S, S) fp32 images, two launches per batch.  Replaces the albumentations pipeline the reference runs on the host cores for
 * every crop (datasets/mp100_cape.py:896-950: Affine, HorizontalFlip, ColorJitter, OneOf(GaussNoise, GaussianBlur, MotionBlur),
 * Resize).  An item is one image and its *plan* (the random numbers of those transforms, drawn by the DataLoader worker:
 * datasets/transforms.py): aug pixel (x, y) samples source pixel (M0 x + M1 y + M2, M3 x + M4 y + M5) bilinearly with zero padding
 * (affine about the centre + flip); colour jitter in `order` (0 brightness, 1 contrast about the image's mean grey, 2 saturation,
 * 3 hue), each clipped to [0, 1]; mode 1 adds N(0, noise_std) from the counter RNG (seed, channel, pixel), mode 2 convolves with
 * the blur_k x blur_k kernel (reflect-101 border); then bilinear resize to S x S (half-pixel centres, edge clamp) and
 * (x - mean[c]) / std[c] when mean != NULL.  `items_dev` is a DEVICE array; `aug` is an (h, w, 3) fp32 workspace per item,
 * `stat` one float per item that the caller zeroes.  max_pixels = the largest h * w of the batch.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* src; float* aug; float* out; float* stat;
  int h, w;
  float M[6];
  int color_on, order[4];
  float bright, contrast, sat, hue;
  int mode;
  float noise_std; uint32_t seed;
  int blur_k; float blur_w[49];
} cape_augment_item;
int cape_augment_batch(const cape_augment_item* items_dev, int n_items, int max_pixels, int out_size, const float* mean,
                       const float* stdv, cape_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPE_HIP_H */
