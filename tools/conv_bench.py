"""Isolated timing of the implicit-GEMM convolution modes (forward im2col, dgrad gather) against the dense product of the same
shape: the 3x3 convolutions of the ResNet trunk at the CAPE batch (32 images, 256 x 256 input)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.hip import ops  # noqa: E402


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    dev = "cuda"
    # (N, H, W, C, O, stride): layer2 / layer3 / layer4 3x3 convolutions
    for (N, H, W, C, O, s) in [(32, 32, 32, 128, 128, 1), (32, 16, 16, 256, 256, 1), (32, 8, 8, 512, 512, 1), (32, 32, 32, 256, 256, 2)]:
        KH = KW = 3
        pad = 1
        OH, OW = (H + 2 * pad - KH) // s + 1, (W + 2 * pad - KW) // s + 1
        M, K = N * OH * OW, KH * KW * C
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(O, KH, KW, C, device=dev)                 # physical layout of a channels_last conv weight
        y = torch.empty(N, OH, OW, O, device=dev)
        dy = torch.randn(N, OH, OW, O, device=dev)
        dx = torch.empty(N, H, W, C, device=dev)
        geom = (N, H, W, C, KH, KW, s, pad, OH, OW, O)
        xd = torch.randn(M, K, device=dev)
        us_f = t(lambda: ops.gemm(x, w, y, M, O, K, a_mode=2, b_mode=0, conv=geom))
        us_d = t(lambda: ops.gemm(xd, w.view(O, K), y.view(M, O), M, O, K))
        fl = 2.0 * M * O * K
        line = f"N={N} {H}x{W} C={C} O={O} s={s}: fwd im2col {us_f:7.1f} us ({fl / us_f / 1e6:6.1f} TF/s) | dense {M}x{O}x{K} {us_d:7.1f} us ({fl / us_d / 1e6:6.1f} TF/s)"
        Md, Kd = N * H * W, KH * KW * O
        us_g = t(lambda: ops.gemm(dy, w, dx, Md, C, Kd, a_mode=3, b_mode=2, conv=geom))
        dyd = torch.randn(Md, Kd, device=dev)
        wt = torch.randn(Kd, C, device=dev)
        us_gd = t(lambda: ops.gemm(dyd, wt, dx.view(Md, C), Md, C, Kd, b_mode=1))
        fl2 = 2.0 * Md * C * Kd
        line += f" || dgrad gather {us_g:7.1f} us ({fl2 / us_g / 1e6:6.1f} TF/s) | dense NN {us_gd:7.1f} us ({fl2 / us_gd / 1e6:6.1f} TF/s)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
