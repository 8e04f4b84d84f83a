"""Isolated timing of the fp32-MFMA GEMM family on the shapes of the CAPE training step (no concurrency)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402

SHAPES = [  # (M, N, K, a_mode, b_mode, split_k)
    (43520, 256, 256, 0, 0, 1), (43520, 256, 1024, 0, 0, 1), (43520, 1024, 256, 0, 0, 1), (43520, 256, 256, 0, 1, 1),
    (43520, 384, 256, 0, 0, 1), (6400, 256, 256, 0, 0, 1), (6400, 1024, 256, 0, 0, 1), (256, 256, 43520, 1, 1, 64),
    (1024, 256, 43520, 1, 1, 16), (256, 1024, 43520, 1, 1, 16), (256, 256, 6400, 1, 1, 24), (8192, 8192, 1024, 0, 0, 1),
    (4096, 4096, 4096, 0, 0, 1),
    # M = 256 x 128: one full round of 128x128 tiles at 2 blocks per CU (tile-efficiency comparison without quantisation)
    (32768, 256, 256, 0, 0, 1), (32768, 1024, 256, 0, 0, 1), (32768, 256, 1024, 0, 0, 1), (32768, 256, 256, 0, 1, 1),
    (1024, 256, 6400, 1, 1, 16), (256, 1024, 6400, 1, 1, 16), (6400, 256, 1024, 0, 0, 1), (6400, 256, 1024, 0, 1, 1),
    (43520, 1024, 256, 0, 1, 1), (6400, 256, 256, 0, 1, 1), (6400, 768, 256, 0, 0, 1), (544, 256, 256, 0, 0, 1), (544, 512, 256, 0, 0, 1),
    (131072, 256, 64, 0, 0, 1), (32768, 512, 128, 0, 0, 1), (43520, 256, 128, 0, 1, 1), (43520, 128, 256, 0, 0, 1),
    (8192, 1024, 256, 0, 0, 1), (2176, 256, 256, 0, 0, 1),
]
if os.environ.get("GEMM_BENCH_WGRAD"):          # k-split sweep of the weight-gradient shapes (pair with CAPE_GEMM_TILE=64|128)
    SHAPES = [(m, n, k, 1, 1, sk) for (m, n, k) in [(256, 256, 43520), (1024, 256, 43520), (256, 1024, 43520), (128, 256, 43520),
                                                     (256, 256, 6400), (1024, 256, 6400), (256, 1024, 6400), (256, 256, 544)]
              for sk in (1, 4, 8, 16, 24, 32, 64, 128) if k // sk >= 128]
if os.environ.get("GEMM_BENCH_SHAPES"):          # "M,N,K,am,bm,sk;M,N,K,am,bm,sk;..."
    SHAPES = [tuple(int(v) for v in item.split(",")) for item in os.environ["GEMM_BENCH_SHAPES"].split(";") if item]
if os.environ.get("GEMM_BENCH_RS_ONLY"):
    SHAPES = [s for s in SHAPES if s[3] == 0 and s[5] == 1 and s[2] in (64, 128, 256)]


def epilogue_sweep():
    """Products with one extra epilogue operand (residual / ReLU-dropout gate from a saved activation / accumulate)."""
    dev = "cuda"
    for (M, N, K, bm) in [(43520, 1024, 256, 1), (43520, 256, 256, 0), (43520, 256, 256, 1), (6400, 1024, 256, 1), (6400, 256, 256, 0)]:
        A = torch.randn(M, K, device=dev)
        B = torch.nn.Parameter(torch.randn((N, K) if bm == 0 else (K, N), device=dev))
        C = torch.zeros(M, N, device=dev)
        X = torch.relu(torch.randn(M, N, device=dev))
        for name, kw in (("plain", {}), ("residual", {"residual": X}), ("gate", {"mask_src": X, "mask_scale": 1.1}), ("accumulate", {"accumulate": True})):
            for _ in range(3):
                ops.gemm(A, B, C, M, N, K, b_mode=bm, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(A, B, C, M, N, K, b_mode=bm, **kw)
            e1.record()
            torch.cuda.synchronize()
            print(f"M={M:6d} N={N:5d} K={K:4d} bm={bm} {name:10s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)


def main():
    if os.environ.get("GEMM_BENCH_EPI"):
        return epilogue_sweep()
    dev = "cuda"
    for (M, N, K, am, bm, sk) in SHAPES:
        A = torch.randn((M, K) if am == 0 else (K, M), device=dev)
        B = torch.randn((N, K) if bm == 0 else (K, N), device=dev)
        C = torch.zeros(M, N, device=dev)
        kw = dict(a_mode=am, b_mode=bm, split_k=sk, accumulate=sk > 1)
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it):
            ops.gemm(A, B, C, M, N, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        line = f"M={M:6d} N={N:5d} K={K:6d} am={am} bm={bm} sk={sk:2d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s"
        if am == 0 and sk == 1 and ops.get_gemm_precision() == "bf16x3" and K in (64, 128, 256):     # packed weight (parameter as B)
            Bp = torch.nn.Parameter(B)
            for _ in range(3):
                ops.gemm(A, Bp, C, M, N, K, **kw)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(it):
                ops.gemm(A, Bp, C, M, N, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            line += f"   | packed {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
