"""Isolated timing of the attention kernels on the decoder self-attention shape of the training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402


def main():
    for N, Lq, Lk, mode in ((32, 200, 200, 1), (32, 200, 17, 2), (16, 17, 17, 2)):
        q, k, v = (torch.randn(N, L, 256, device="cuda") for L in (Lq, Lk, Lk))
        kpm = torch.zeros(N, Lk, dtype=torch.uint8, device="cuda") if mode == 2 else None
        go = torch.randn(N, Lq, 256, device="cuda")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        def fwd():
            return ops.attn_fwd(q, k, v, N, 8, Lq, Lk, 32 ** -0.5, mask_mode=mode, kpm=kpm)
        O, lse = fwd()
        def bwd():
            ops.attn_bwd(go, q, k, v, O, lse, dq, dk, dv, N, 8, Lq, Lk, 32 ** -0.5, mask_mode=mode, kpm=kpm)
        cases = [("fwd", fwd), ("bwd", bwd)]
        if ops.attn_mm_ok(N, 8, Lq, Lk):
            O2, P, Pu = ops.attn_mm_fwd(q, k, v, N, 8, Lq, Lk, 32 ** -0.5, mask_mode=mode, kpm=kpm)
            cases += [("mm fwd", lambda: ops.attn_mm_fwd(q, k, v, N, 8, Lq, Lk, 32 ** -0.5, mask_mode=mode, kpm=kpm)),
                      ("mm bwd", lambda: ops.attn_mm_bwd(go, q, k, v, P, Pu, dq, dk, dv, N, 8, Lq, Lk, 32 ** -0.5))]
        for name, fn in cases:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"N={N} Lq={Lq} Lk={Lk} mode={mode} {name}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
