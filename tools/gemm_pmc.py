"""A handful of GEMM launches for counter collection (rocprofv3 --pmc): shapes from the CAPE step, 3 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402

SHAPES = [(43520, 256, 256, 0, 0, 1), (43520, 1024, 256, 0, 0, 1), (43520, 256, 256, 0, 1, 1), (256, 256, 43520, 1, 1, 64),
          (4096, 4096, 4096, 0, 0, 1)]

for (M, N, K, am, bm, sk) in SHAPES:
    A = torch.randn((M, K) if am == 0 else (K, M), device="cuda")
    B = torch.randn((N, K) if bm == 0 else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, a_mode=am, b_mode=bm, split_k=sk, accumulate=sk > 1)
    torch.cuda.synchronize()
