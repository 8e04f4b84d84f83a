// Lab harness for the register-stationary GEMM (csrc/gemm_rs.hip): stand-alone timing + per-block phase stamps, no torch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRS_STAMPS] tools/lab/rs_lab.hip -o tools/lab/rs_lab
// run:   tools/lab/rs_lab [M N K b_mode]
#include "../../category-agnostic-pose-estimation_amd/csrc/error.hip"
#include "../../category-agnostic-pose-estimation_amd/csrc/gemm_rs.hip"
#include <vector>
#include <algorithm>
#include <math.h>

__global__ void fill(float* p, long long n, unsigned seed) {
  long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = ((x & 0xFFFF) / 32768.f - 1.f); }
}
__global__ void ref_check(const float* A, const float* B, const float* C, int M, int N, int K, int bm, float* maxerr) {
  int row = blockIdx.x * 37 % M, col = threadIdx.x % N;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[(long long)row * K + k] * (bm == 0 ? B[(long long)col * K + k] : B[(long long)k * N + col]);
  float e = fabsf((float)s - C[(long long)row * N + col]);
  atomicMax((int*)maxerr, __float_as_int(e));
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 43520, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 256;
  int bm = argc > 4 ? atoi(argv[4]) : 0;
  int gen = argc > 5 ? atoi(argv[5]) : 0;   // 1: residual + relu, 2: gate (mask_src), 3: accumulate
  float *A, *B, *C, *err;
  hipMalloc(&A, 4ll * M * K); hipMalloc(&B, 4ll * N * K); hipMalloc(&C, 4ll * M * N); hipMalloc(&err, 4);
  fill<<<(unsigned)(((long long)M * K + 255) / 256), 256>>>(A, (long long)M * K, 1);
  fill<<<(N * K + 255) / 256, 256>>>(B, (long long)N * K, 2);
  hipMemset(err, 0, 4);
  GemmP p = {};
  p.M = M; p.N = N; p.K = K; p.A = A; p.lda = K; p.B = B; p.ldb = bm == 0 ? K : N; p.C = C; p.ldc = N; p.split_k = 1;
  float* R = nullptr;
  if (gen) { hipMalloc(&R, 4ll * M * N); hipMemset(R, 0, 4ll * M * N); }
  if (gen == 1) { p.residual = R; p.ldr = N; p.relu = 0; }
  if (gen == 2) { fill<<<(unsigned)(((long long)M * N + 255) / 256), 256>>>(R, (long long)M * N, 3); p.mask_src = R; p.ldm = N; p.mask_scale = 1.f; }
  if (gen == 3) { p.accumulate = 1; }
  if (gen == 4) { uint64_t* st8; hipMalloc(&st8, 16); uint64_t hs[2] = {1234, 5}; hipMemcpy(st8, hs, 16, hipMemcpyHostToDevice); p.rng_state = st8; p.rng_stream = 3; p.drop_thresh = cape_drop_threshold(0.1f); p.inv_keep = 1.f / 0.9f; p.relu = 1; }
  if (getenv("RS_LAB_PACKED")) {
    unsigned short* pk; hipMalloc(&pk, cape_packed_weight_bytes(N, K));
    cape_pack_item it = {B, pk, (long long)(bm == 0 ? K : N), N, K, bm, 0};
    cape_pack_item* itd; hipMalloc(&itd, sizeof(it)); hipMemcpy(itd, &it, sizeof(it), hipMemcpyHostToDevice);
    if (cape_pack_weights(itd, 1, 16, 0)) { printf("pack error: %s\n", cape_last_error()); return 1; }
    p.Bpack = pk;
  }
  if (!cape_gemm_rs_eligible(p, 0, bm)) { printf("not eligible\n"); return 1; }
#ifdef RS_STAMPS
  long long* st; hipMalloc(&st, 8 * 16 * 4096); hipMemset(st, 0, 8 * 16 * 4096);
  hipMemcpyToSymbol(HIP_SYMBOL(g_rs_stamps), &st, sizeof(st));
#endif
  for (int i = 0; i < 3; ++i) if (cape_gemm_rs_launch(p, bm, 0)) { printf("launch error: %s\n", cape_last_error()); return 1; }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) cape_gemm_rs_launch(p, bm, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ref_check<<<64, 256>>>(A, B, C, M, N, K, bm, err);
  float herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  printf("M=%d N=%d K=%d bm=%d gen=%d: %.1f us/launch  %.1f TF/s  %.2f TB/s(A+C)  maxerr %.2e\n", M, N, K, bm, gen, ms / it * 1e3,
         2.0 * M * N * K / (ms / it * 1e-3) / 1e12, 4.0 * ((double)M * K + (double)M * N) / (ms / it * 1e-3) / 1e12, herr);
#ifdef RS_STAMPS
  std::vector<long long> h(16 * 4096);
  hipMemcpy(h.data(), st, 8 * 16 * 4096, hipMemcpyDeviceToHost);
  int nb = 0; while (nb < 4096 && h[nb * 16]) ++nb;
  long long t0 = h[0]; for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[b * 16]);
  printf("blocks %d; stamps (cycles from the earliest block start): start | B done | first sync | unit ends...\n", nb);
  for (int b : {0, 1, 8, nb / 2, nb - 1}) {
    printf("  block %4d:", b);
    for (int i = 0; i < 8 && h[b * 16 + i]; ++i) printf(" %8lld", h[b * 16 + i] - t0);
    printf("\n");
  }
  // averages of phase durations
  double d[8] = {0}; int cnt[8] = {0};
  for (int b = 0; b < nb; ++b) for (int i = 1; i < 8 && h[b * 16 + i]; ++i) { d[i] += h[b * 16 + i] - h[b * 16 + i - 1]; cnt[i]++; }
  printf("  mean phase cycles:"); for (int i = 1; i < 8 && cnt[i]; ++i) printf(" %.0f", d[i] / cnt[i]); printf("\n");
  // second unit of a block, wave 0: start of half 0 MFMAs | its MFMAs issued | (unused) | half 1 start | issued | (unused) | stores issued | barrier passed
  double f[8] = {0}; int fc = 0;
  for (int b = 0; b < nb; ++b) if (h[b * 16 + 8] && h[b * 16 + 15]) { for (int i = 0; i < 8; ++i) f[i] += (double)(h[b * 16 + 8 + i] - h[b * 16 + 8]); ++fc; }
  if (fc) { printf("  second unit, wave 0, cycles from its first MFMA: h0 issued %.0f | h1 start %.0f | h1 issued %.0f | stores issued %.0f | barrier %.0f  (%d blocks)\n",
                   f[1] / fc, f[3] / fc, f[4] / fc, f[6] / fc, f[7] / fc, fc); }
#endif
  return 0;
}
