"""Isolated timing of the geometric support encoder (forward + backward, training mode) at the headline batch: 32 graphs x 17
keypoints, GCN pre-encoder on."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import functional as HF  # noqa: E402
from cape_amd.models.geometric_support_encoder import GeometricSupportEncoder  # noqa: E402


def main():
    B, P = int(os.environ.get("GRAPHS", "32")), int(os.environ.get("POINTS", "17"))
    torch.manual_seed(0)
    enc = GeometricSupportEncoder(use_gcn_preenc=True).cuda().train()
    coords = torch.rand(B, P, 2, device="cuda")
    mask = torch.zeros(B, P, dtype=torch.bool, device="cuda")
    mask[:, 14:] = True
    sk = [[[i, i + 1] for i in range(13)] for _ in range(B)]
    HF.Runtime.seed(1, "cuda")

    def step():
        out = enc(coords, mask, sk)
        out.backward(torch.ones_like(out))
        HF.Runtime.join()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"support encoder fwd+bwd, {B} graphs x {P} points: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us GPU, "
          f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms wall")
    with torch.no_grad():
        enc.eval()
        for _ in range(3):
            enc(coords, mask, sk)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            enc(coords, mask, sk)
        e1.record()
        torch.cuda.synchronize()
        print(f"support encoder forward (eval): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")


if __name__ == "__main__":
    main()
