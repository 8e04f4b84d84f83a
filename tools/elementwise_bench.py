"""Isolated timing of the HBM-bound row kernels at the encoder's size (43520 x 256 fp32 = 44.5 MB per tensor)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = "cuda"
    rows, C = 43520, 256
    mb = rows * C * 4 / 1e6
    x, y, d = (torch.randn(rows, C, device=dev) for _ in range(3))
    pos = torch.randn(rows, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rng = ops.RngState(1, dev)
    for p in (0.0, 0.1):
        out, mean, rstd, _ = ops.add_layernorm_fwd(x, y, g, b, dropout_p=p, rng=rng, rng_stream=3)
        us = t(lambda: ops.add_layernorm_fwd(x, y, g, b, dropout_p=p, rng=rng, rng_stream=3))
        print(f"add_ln_fwd  p={p}: {us:6.1f} us  ({3 * mb / us * 1e-3:5.2f} (/1e3) TB/s for 3 passes)")
        us = t(lambda: ops.add_layernorm_fwd(x, y, g, b, pos=pos, dropout_p=p, rng=rng, rng_stream=3))
        print(f"add_ln_fwd+pos p={p}: {us:6.1f} us  ({5 * mb / us * 1e-3:5.2f} (/1e3) TB/s for 5 passes)")
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        us = t(lambda: ops.add_layernorm_bwd(d, None, x, y, g, mean, rstd, dg, db, dropout_p=p, rng=rng, rng_stream=3))
        print(f"add_ln_bwd  p={p}: {us:6.1f} us  ({(5 if p else 4) * mb / us * 1e-3:5.2f} (/1e3) TB/s for {5 if p else 4} passes)")
    us = t(lambda: ops.add_n([x, y, d]))
    print(f"add_n(3)        : {us:6.1f} us  ({4 * mb / us * 1e-3:5.2f} (/1e3) TB/s for 4 passes)")
    us = t(lambda: ops.add_n([x, y]))
    print(f"add_n(2)        : {us:6.1f} us  ({3 * mb / us * 1e-3:5.2f} (/1e3) TB/s for 3 passes)")
    # ResNet layer2 output (32, 32, 32, 512): folded-BN / ReLU backward, in-place affine pass
    a = torch.randn(32, 32, 32, 512, device=dev); yb = torch.relu(torch.randn_like(a)); sc = torch.rand(512, device=dev) + 0.5
    mb2 = a.numel() * 4 / 1e6
    us = t(lambda: ops.bn_relu_bwd(a, yb, sc, True, True))
    print(f"bn_relu_bwd (+res) 67 MB: {us:6.1f} us  ({4 * mb2 / us:5.2f} GB/ms for 4 passes)")
    us = t(lambda: ops.bn_relu_bwd(a, yb, sc, True, False))
    print(f"bn_relu_bwd        67 MB: {us:6.1f} us  ({3 * mb2 / us:5.2f} GB/ms for 3 passes)")
    us = t(lambda: ops.affine_act_(a, sc, sc, yb, True))
    print(f"affine_act_ (+res) 67 MB: {us:6.1f} us  ({3 * mb2 / us:5.2f} GB/ms for 3 passes)")
    us = t(lambda: ops.relu_drop_bwd(a, yb, 1.0))
    print(f"relu_drop_bwd      67 MB: {us:6.1f} us  ({3 * mb2 / us:5.2f} GB/ms for 3 passes)")
    # gradient norm over one 23 M-float arena
    gflat = torch.randn(23_000_000, device=dev); acc = torch.zeros(256, device=dev)
    us = t(lambda: ops.sumsq(gflat, acc))
    print(f"sumsq 92 MB             : {us:6.1f} us  ({92.0 / us:5.2f} GB/ms)")
    # GroupNorm of input_proj level 0: N = 32, HW = 1024, C = 256
    xg = torch.randn(32, 1024, 256, device=dev); og = torch.empty(32, 1024, 256, device=dev)
    mean, rstd = ops.groupnorm_fwd(xg, g, b, og, 1024 * 256, 32, 1024, 256)
    us = t(lambda: ops.groupnorm_fwd(xg, g, b, og, 1024 * 256, 32, 1024, 256))
    print(f"groupnorm_fwd 33.5 MB   : {us:6.1f} us  ({3 * 33.5 / us:5.2f} GB/ms for 3 passes)")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    us = t(lambda: ops.groupnorm_bwd(og, 1024 * 256, xg, g, mean, rstd, dg, db, 32, 1024, 256))
    print(f"groupnorm_bwd 33.5 MB   : {us:6.1f} us  ({5 * 33.5 / us:5.2f} GB/ms for 5 passes)")


if __name__ == "__main__":
    main()
