// support.hip -- geometric support encoder pieces that are not plain GEMM / LayerNorm / attention:
// coordinate embedding + sine position encodings, adjacency construction, GCN aggregation.
// The graphs are tiny (P <= 100 keypoints x 256 channels): one block per graph, adjacency in LDS.
#include "common.h"

namespace {

constexpr int MAXP = 100;   // PositionalEncoding1D max_len (geometric_support_encoder.py:98-102)

// h = relu(coords @ W0^T + b0) ; pe = [sine(y) | sine(x)] + pe1d[p]
__global__ void support_embed_fwd_kernel(const float* coords, const float* W0, const float* b0, const float* pe1d,
                                         const float* dim_t, float* h, float* pe, int P, int C) {
  const long long r = blockIdx.x;        // row = n*P + p
  const int p = (int)(r % P);
  const float x = coords[r * 2 + 0], y = coords[r * 2 + 1];
  const int half = C >> 1;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float v = x * W0[c * 2 + 0] + y * W0[c * 2 + 1] + b0[c];
    h[r * C + c] = fmaxf(v, 0.f);
    const int k = c < half ? c : c - half;
    const float e = (c < half ? y : x) * 6.28318530718f / dim_t[k];
    pe[r * C + c] = ((k & 1) ? cosf(e) : sinf(e)) + pe1d[p * C + c];
  }
}

// dW0 (C, 2) += sum_r g[r][c] * coords[r], db0 += sum_r g[r][c], g = d_h gated by relu.  Rows are split over blockIdx.y
// (32 rows per block) and over the 4 waves of a block (lane = channel: coalesced 256-byte row reads); partial sums meet in
// LDS and leave as three float atomics per channel and block (R / 32 adders per address).
constexpr int SEB_ROWS = 32;
__global__ void __launch_bounds__(256) support_embed_bwd_kernel(const float* d_h, const float* h, const float* coords, float* dW0,
                                                                float* db0, long long R, int C) {
  __shared__ float part[4][3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * SEB_ROWS, r1 = min(R, r0 + SEB_ROWS);
  float gx = 0.f, gy = 0.f, gb = 0.f;
  if (c < C)
    for (long long r = r0 + w; r < r1; r += 4) {
      const float g = h[r * C + c] > 0.f ? d_h[r * C + c] : 0.f;
      gx += g * coords[r * 2 + 0];
      gy += g * coords[r * 2 + 1];
      gb += g;
    }
  part[w][0][lane] = gx; part[w][1][lane] = gy; part[w][2][lane] = gb;
  __syncthreads();
  if (w == 0 && c < C) {
    atomicAdd(&dW0[c * 2 + 0], part[0][0][lane] + part[1][0][lane] + part[2][0][lane] + part[3][0][lane]);
    atomicAdd(&dW0[c * 2 + 1], part[0][1][lane] + part[1][1][lane] + part[2][1][lane] + part[3][1][lane]);
    atomicAdd(&db0[c], part[0][2][lane] + part[1][2][lane] + part[2][2][lane] + part[3][2][lane]);
  }
}

__global__ void adjacency_kernel(const int* edges, const int* edge_start, const uint8_t* mask, float* adj, int P) {
  extern __shared__ float a[];           // P*P
  const int n = blockIdx.x;
  for (int i = threadIdx.x; i < P * P; i += blockDim.x) a[i] = 0.f;
  __syncthreads();
  for (int e = edge_start[n] + threadIdx.x; e < edge_start[n + 1]; e += blockDim.x) {
    int i = edges[2 * e], j = edges[2 * e + 1];
    if (i < P && j < P) {                // graph_utils.py:59-60 ; negative indices wrap like torch indexing
      if (i < 0) i += P;
      if (j < 0) j += P;
      if (i >= 0 && j >= 0) a[i * P + j] = 1.f;
    }
  }
  __syncthreads();
  const uint8_t* m = mask + (long long)n * P;
  float* o0 = adj + (long long)n * 2 * P * P;
  float* o1 = o0 + P * P;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const float ki = m[i] ? 0.f : 1.f;
    float rs = 0.f;
    for (int j = 0; j < P; ++j) {
      const float v = fmaxf(a[i * P + j], a[j * P + i]) * ki * (m[j] ? 0.f : 1.f);
      rs += v;
    }
    for (int j = 0; j < P; ++j) {
      const float v = fmaxf(a[i * P + j], a[j * P + i]) * ki * (m[j] ? 0.f : 1.f);
      o1[i * P + j] = rs > 0.f ? v / rs : 0.f;         // nan_to_num(0/0) = 0
      o0[i * P + j] = (i == j) ? ki : 0.f;
    }
  }
}

// out[n,w,c] = relu( adj0[w,w]*y[n,w,c] + sum_v adj1[v,w]*y[n,v,C+c] )
// one block per (graph, node), one thread per channel: the first version ran one block per graph (32 blocks on 256 CUs) with
// every thread walking all P x P pairs (27 / 43 us for 544 x 256 outputs)
__global__ void gcn_fwd_kernel(const float* __restrict__ y, const float* __restrict__ adj, float* __restrict__ out, int P, int C) {
  const int n = blockIdx.x / P, w = blockIdx.x - n * P;
  const float* a = adj + (long long)n * 2 * P * P;
  const float* yn = y + (long long)n * P * 2 * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = a[w * P + w] * yn[(long long)w * 2 * C + c];
    for (int v = 0; v < P; ++v) s += a[P * P + v * P + w] * yn[(long long)v * 2 * C + C + c];
    out[((long long)n * P + w) * C + c] = fmaxf(s, 0.f);
  }
}

__global__ void gcn_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ out, const float* __restrict__ adj,
                               float* __restrict__ d_y, int P, int C) {
  const int n = blockIdx.x / P, v = blockIdx.x - n * P;
  const float* a = adj + (long long)n * 2 * P * P;
  const float* gn = d_out + (long long)n * P * C;
  const float* on = out + (long long)n * P * C;
  float* dyn = d_y + (long long)n * P * 2 * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float gv = on[(long long)v * C + c] > 0.f ? gn[(long long)v * C + c] : 0.f;
    dyn[(long long)v * 2 * C + c] = a[v * P + v] * gv;
    float s = 0.f;
    for (int w = 0; w < P; ++w) {
      const float gw = on[(long long)w * C + c] > 0.f ? gn[(long long)w * C + c] : 0.f;
      s += a[P * P + v * P + w] * gw;
    }
    dyn[(long long)v * 2 * C + C + c] = s;
  }
}

// key-padding mask glue of the support encoder (geometric_support_encoder.py:201-220) in one launch: one wave per graph.
//   mask (N, P) u8, non-zero = ignore.  A graph whose keypoints are ALL masked would give softmax rows of -inf: keypoint 0 is
//   unmasked for the attention (kpm) and the graph's output rows are zeroed afterwards (zero).
//   kpm[n][p]  = mask[n][p], except kpm[n][0] = 0 when every keypoint of graph n is masked
//   zero[n][p] = all-masked(n)  (| mask[n][p] when pad_rows: the nested-tensor fast path of nn.TransformerEncoder returns 0 at
//                padded positions)
__global__ void __launch_bounds__(64) support_masks_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ kpm,
                                                           uint8_t* __restrict__ zero, int N, int P, int pad_rows) {
  const int n = blockIdx.x, lane = threadIdx.x;
  bool any_valid = false;
  for (int p_ = lane; p_ < P; p_ += 64) any_valid = any_valid || (mask[(long long)n * P + p_] == 0);
  const bool all_masked = __ballot(any_valid) == 0ull;
  for (int p_ = lane; p_ < P; p_ += 64) {
    const uint8_t m = mask[(long long)n * P + p_] != 0;
    kpm[(long long)n * P + p_] = (p_ == 0 && all_masked) ? 0 : m;
    zero[(long long)n * P + p_] = (all_masked || (pad_rows && m)) ? 1 : 0;
  }
}

}  // namespace

extern "C" int cape_support_embed_fwd(const float* coords, const float* W0, const float* b0, const float* pe1d,
                                      const float* dim_t, float* h, float* pe, int N, int P, int C, cape_stream_t stream) {
  CAPE_REQUIRE(coords && W0 && b0 && pe1d && dim_t && h && pe, "cape_support_embed_fwd: null pointer");
  CAPE_REQUIRE(C == 256 && P >= 1 && P <= MAXP, "cape_support_embed_fwd: C must be 256 and 1 <= P <= %d", MAXP);
  if (N <= 0) return 0;
  hipLaunchKernelGGL(support_embed_fwd_kernel, dim3((unsigned)((long long)N * P)), dim3(256), 0, as_stream(stream), coords,
                     W0, b0, pe1d, dim_t, h, pe, P, C);
  CAPE_LAUNCH_CHECK("cape_support_embed_fwd");
  return 0;
}

extern "C" int cape_support_embed_bwd(const float* d_h, const float* h, const float* coords, float* dW0, float* db0, int N,
                                      int P, int C, cape_stream_t stream) {
  CAPE_REQUIRE(d_h && h && coords && dW0 && db0, "cape_support_embed_bwd: null pointer");
  if (N <= 0) return 0;
  const long long R = (long long)N * P;
  CAPE_REQUIRE((R + SEB_ROWS - 1) / SEB_ROWS <= 65535, "cape_support_embed_bwd: too many rows");
  hipLaunchKernelGGL(support_embed_bwd_kernel, dim3((C + 63) / 64, (unsigned)((R + SEB_ROWS - 1) / SEB_ROWS)), dim3(256), 0,
                     as_stream(stream), d_h, h, coords, dW0, db0, R, C);
  CAPE_LAUNCH_CHECK("cape_support_embed_bwd");
  return 0;
}

extern "C" int cape_adjacency(const int* edges, const int* edge_start, const uint8_t* mask, float* adj, int N, int P,
                              cape_stream_t stream) {
  CAPE_REQUIRE(edge_start && mask && adj && P >= 1 && P <= MAXP, "cape_adjacency: bad arguments");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(adjacency_kernel, dim3(N), dim3(128), sizeof(float) * P * P, as_stream(stream), edges, edge_start, mask,
                     adj, P);
  CAPE_LAUNCH_CHECK("cape_adjacency");
  return 0;
}

extern "C" int cape_gcn_aggregate_fwd(const float* y, const float* adj, float* out, int N, int P, int C, cape_stream_t stream) {
  CAPE_REQUIRE(y && adj && out && P >= 1 && P <= MAXP, "cape_gcn_aggregate_fwd: bad arguments");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(gcn_fwd_kernel, dim3((unsigned)((long long)N * P)), dim3(256), 0, as_stream(stream), y, adj, out, P, C);
  CAPE_LAUNCH_CHECK("cape_gcn_aggregate_fwd");
  return 0;
}

extern "C" int cape_gcn_aggregate_bwd(const float* d_out, const float* out, const float* adj, float* d_y, int N, int P, int C,
                                      cape_stream_t stream) {
  CAPE_REQUIRE(d_out && out && adj && d_y && P >= 1 && P <= MAXP, "cape_gcn_aggregate_bwd: bad arguments");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(gcn_bwd_kernel, dim3((unsigned)((long long)N * P)), dim3(256), 0, as_stream(stream), d_out, out, adj, d_y, P, C);
  CAPE_LAUNCH_CHECK("cape_gcn_aggregate_bwd");
  return 0;
}

extern "C" int cape_support_masks(const uint8_t* mask, uint8_t* kpm, uint8_t* zero, int N, int P, int pad_rows, cape_stream_t stream) {
  CAPE_REQUIRE(mask && kpm && zero && N >= 0 && P > 0, "cape_support_masks: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(support_masks_kernel, dim3((unsigned)N), dim3(64), 0, as_stream(stream), mask, kpm, zero, N, P, pad_rows);
  CAPE_LAUNCH_CHECK("cape_support_masks");
  return 0;
}
