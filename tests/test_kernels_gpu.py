"""GPU parity tests of every HIP kernel behind the C ABI against CPU restatements
(oracle/cape_ref.py where the op exists there, plain torch fp32 CPU math otherwise).
Tolerances: fp32 arithmetic with a different summation order -> 1e-4 relative to the tensor scale
unless stated; integer/index outputs exact."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import cape_amd  # noqa: F401
    from cape_amd.hip import ops
from oracle import cape_ref

DEV = "cuda"
CFG = cape_ref.Cfg()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def close(got, ref, tol=1e-4, name=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = max(1.0, ref.abs().max().item() if ref.numel() else 1.0)
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 70, 256), (128, 128, 32), (1000, 256, 1024), (640, 3, 256), (64, 384, 256),
                                   (50, 256, 2), (33, 2, 36), (5000, 1024, 256), (17, 256, 256)])
def test_gemm_nt_epilogues(M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, sc = rnd(M, N, seed=4), rnd(N, seed=5).abs() + 0.5
    xd, wd, bd, rd, sd = (t.to(DEV) for t in (x, w, b, res, sc))
    out = torch.empty(M, N, device=DEV)
    ops.gemm(xd, wd, out, M, N, K, bias=bd)
    close(out, F.linear(x, w, b), name="bias")
    ops.gemm(xd, wd, out, M, N, K, bias=bd, scale=sd, residual=rd, relu=True)
    close(out, F.relu(x @ w.t() * sc + b + res), name="scale+bias+res+relu")
    out2 = out.clone()
    ops.gemm(xd, wd, out2, M, N, K, accumulate=True)
    close(out2, out.cpu() + x @ w.t(), name="accumulate")
    if K >= 512:
        out3 = torch.zeros(M, N, device=DEV)
        ops.gemm(xd, wd, out3, M, N, K, split_k=4)
        close(out3, x @ w.t(), name="split_k")


@pytest.mark.parametrize("M,N,K,bm", [(43520, 256, 256, 0), (6400, 256, 256, 1), (6401, 384, 256, 0), (130, 1024, 256, 1),
                                      (65, 32, 64, 0), (8191, 512, 128, 0), (1000, 100, 128, 1), (544, 256, 256, 0),
                                      (4097, 768, 256, 0), (129, 36, 64, 1)])
def test_gemm_rs_register_stationary(M, N, K, bm):
    """Dense A against a <= 256-deep weight takes the register-stationary kernel (gemm_rs.hip) in bf16x3 mode: every
    epilogue, strided operands, ragged M / N, both weight layouts; compared with an fp64 product."""
    if ops.get_gemm_precision() != "bf16x3":
        pytest.skip("the register-stationary kernel is the bf16x3 path")
    x = rnd(M, K + 8, seed=1)[:, :K]                                   # lda = K + 8
    w = rnd(N, K, seed=2, scale=K ** -0.5) if bm == 0 else rnd(K, N, seed=2, scale=K ** -0.5)
    b, sc, res = rnd(N, seed=3), rnd(N, seed=5).abs() + 0.5, rnd(M, N, seed=4)
    ref = (x.double() @ (w.double().t() if bm == 0 else w.double()))
    xd_full = rnd(M, K + 8, seed=1).to(DEV)
    xd = xd_full[:, :K]
    wd = w.to(DEV)
    out = torch.full((M, N + 4), 3.0, device=DEV)                      # ldc = N + 4: the pad columns must stay untouched
    ops.gemm(xd, wd, out, M, N, K, b_mode=bm, lda=K + 8, ldc=N + 4, bias=b.to(DEV))
    close(out[:, :N], (ref + b).float(), tol=3e-5, name="rs bias")
    assert float((out[:, N:] - 3.0).abs().sum()) == 0
    o2 = torch.empty(M, N, device=DEV)
    ops.gemm(xd, wd, o2, M, N, K, b_mode=bm, lda=K + 8, bias=b.to(DEV), scale=sc.to(DEV), residual=res.to(DEV), relu=True)
    want = F.relu(ref * sc + b + res).float()
    close(o2, want, tol=3e-5, name="rs scale+bias+res+relu")
    o3 = o2.clone()
    ops.gemm(xd, wd, o3, M, N, K, b_mode=bm, lda=K + 8, accumulate=True)
    close(o3, want + ref.float(), tol=5e-5, name="rs accumulate")
    gate = (rnd(M, N, seed=9) > 0).float()
    o4 = torch.empty(M, N, device=DEV)
    ops.gemm(xd, wd, o4, M, N, K, b_mode=bm, lda=K + 8, mask_src=gate.to(DEV), mask_scale=1.25)
    close(o4, (ref * gate * 1.25).float(), tol=3e-5, name="rs gate")
    # dropout epilogue: same counter indexing as the tiled kernel -> identical masks
    rng = ops.RngState(77, DEV)
    o5, o6 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm(xd, wd, o5, M, N, K, b_mode=bm, lda=K + 8, relu=True, dropout_p=0.1, rng=rng, rng_stream=5)
    if K == 256 and N % 4 == 0:                                        # the tiled kernel via a split K that it alone serves
        xa = torch.cat([xd, torch.zeros(M, 32, device=DEV)], 1).contiguous()
        wa = torch.cat([wd, torch.zeros(N, 32, device=DEV)], 1).contiguous() if bm == 0 else torch.cat([wd, torch.zeros(32, N, device=DEV)], 0).contiguous()
        ops.gemm(xa, wa, o6, M, N, K + 32, b_mode=bm, relu=True, dropout_p=0.1, rng=rng, rng_stream=5)
        assert bool(((o5 == 0) == (o6 == 0)).all()), "dropout masks differ between the two kernels"
        close(o5, o6, tol=3e-5, name="rs dropout values")
    keep = (o5 != 0).float().mean().item()
    assert 0.3 < keep < 0.6                                            # relu keeps ~half, dropout 90 % of those


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (32, 256, 256), (64, 1024, 256), (32, 256, 1024), (7, 3, 36), (33, 70, 516)])
def test_gemm_skinny_rows(M, N, K):
    """M <= 64 dense NT products take the exact-fp32 FMA kernel (the cached decode step): epilogues and strided outputs."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, sc = rnd(M, N, seed=4), rnd(N, seed=5).abs() + 0.5
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(xd, wd, out, M, N, K, bias=b.to(DEV), scale=sc.to(DEV), residual=res.to(DEV), relu=True)
    close(out, F.relu(x @ w.t() * sc + b + res), tol=2e-6, name="skinny epilogue")
    big = torch.full((M, 5, N), 7.0, device=DEV)                   # row `2` of a (M, 5, N) cache: ldc = 5 * N
    ops.gemm(xd, wd, big[:, 2], M, N, K, bias=b.to(DEV), ldc=5 * N)
    close(big[:, 2], F.linear(x, w, b), tol=2e-6, name="skinny strided out")
    assert float((big[:, 1] - 7).abs().sum()) == 0 and float((big[:, 3] - 7).abs().sum()) == 0
    ops.gemm(xd, wd, out, M, N, K, accumulate=True)
    close(out, F.relu(x @ w.t() * sc + b + res) + x @ w.t(), tol=4e-6, name="skinny accumulate")


@pytest.mark.parametrize("M,N,K", [(300, 256, 70), (640, 256, 3), (1000, 1024, 256), (77, 36, 2)])
def test_gemm_nn_dgrad(M, N, K):
    dy, w = rnd(M, K, seed=1), rnd(K, N, seed=2)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(dy.to(DEV), w.to(DEV), out, M, N, K, a_mode=0, b_mode=1)
    close(out, dy @ w, name="nn")


@pytest.mark.parametrize("Mo,Ni,Kr", [(70, 256, 300), (3, 256, 640), (1024, 256, 3000), (256, 2, 100)])
def test_gemm_tn_wgrad(Mo, Ni, Kr):
    dy, x = rnd(Kr, Mo, seed=1), rnd(Kr, Ni, seed=2)
    out = torch.zeros(Mo, Ni, device=DEV)
    ops.gemm(dy.to(DEV), x.to(DEV), out, Mo, Ni, Kr, a_mode=1, b_mode=1, accumulate=True, split_k=ops.pick_split_k(Mo, Ni, Kr))
    close(out, dy.t() @ x, tol=2e-4, name="tn")


def _conv_case(N, H, W, C, O, k, stride, pad, seed=0):
    x = rnd(N, C, H, W, seed=seed)
    w = rnd(O, C, k, k, seed=seed + 1, scale=(C * k * k) ** -0.5)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    geom = (N, H, W, C, k, k, stride, pad, OH, OW, O)
    return x, w, geom, OH, OW


@pytest.mark.parametrize("N,H,W,C,O,k,stride,pad", [(2, 16, 16, 64, 64, 3, 1, 1), (2, 17, 15, 32, 48, 3, 2, 1),
                                                    (3, 16, 16, 64, 128, 1, 2, 0), (2, 32, 32, 4, 64, 7, 2, 3),
                                                    (2, 8, 8, 128, 256, 1, 1, 0), (1, 8, 8, 256, 256, 3, 2, 1)])
def test_conv_fwd_dgrad_wgrad(N, H, W, C, O, k, stride, pad):
    x, w, geom, OH, OW = _conv_case(N, H, W, C, O, k, stride, pad)
    sc, sh = rnd(O, seed=7).abs() + 0.5, rnd(O, seed=8)
    x.requires_grad_(True); w.requires_grad_(True)
    y_ref = F.conv2d(x, w, stride=stride, padding=pad)
    z_ref = F.relu(y_ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    g = rnd(*y_ref.shape, seed=9)
    y_ref.backward(g)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)                 # NHWC
    wn = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV)                 # (O,KH,KW,C)
    M, K = N * OH * OW, k * k * C
    out = torch.empty(M, O, device=DEV)
    ops.gemm(xn, wn, out, M, O, K, a_mode=2, b_mode=0, conv=geom, scale=sc.to(DEV), bias=sh.to(DEV), relu=True)
    close(out.view(N, OH, OW, O).permute(0, 3, 1, 2), z_ref, name="conv fwd")
    gn = g.permute(0, 2, 3, 1).contiguous().to(DEV)
    # dgrad
    dx = torch.empty(N * H * W, C, device=DEV)
    ops.gemm(gn, wn, dx, N * H * W, C, k * k * O, a_mode=3, b_mode=2, conv=geom)
    close(dx.view(N, H, W, C).permute(0, 3, 1, 2), x.grad, tol=2e-4, name="conv dgrad")
    # wgrad
    dw = torch.zeros(O, k * k * C, device=DEV)
    ops.gemm(gn, xn, dw, O, k * k * C, M, a_mode=1, b_mode=3, lda=O, conv=geom, accumulate=True,
             split_k=ops.pick_split_k(O, k * k * C, M))
    close(dw.view(O, k, k, C).permute(0, 3, 1, 2), w.grad, tol=2e-4, name="conv wgrad")


@pytest.mark.parametrize("N,H,W,C,O,k,stride,pad,res", [(2, 8, 8, 512, 512, 3, 1, 1, False), (2, 8, 8, 2048, 256, 3, 2, 1, False),
                                                       (8, 4, 4, 2048, 512, 1, 1, 0, True)])
def test_conv_node_k_split_path(N, H, W, C, O, k, stride, pad, res):
    """HF.conv_bn_act in grad mode takes the atomic k-split for few-tile / deep-contraction layers (zero fill, bias with split 0,
    FrozenBN / ReLU / shortcut in a separate in-place pass; dgrad k-split too): same values as the single-pass launch that
    inference uses, and the same gradients as torch (ref backbone.py:28-35 FrozenBatchNorm2d folding, torchvision Bottleneck)."""
    from cape_amd.hip import functional as HF
    x, w, geom, OH, OW = _conv_case(N, H, W, C, O, k, stride, pad)
    sc, sh = rnd(O, seed=7).abs() + 0.5, rnd(O, seed=8)
    r = rnd(N, O, OH, OW, seed=10) if res else None
    x.requires_grad_(True); w.requires_grad_(True)
    pre = F.conv2d(x, w, stride=stride, padding=pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    z_ref = F.relu(pre + r if res else pre)
    g = rnd(*z_ref.shape, seed=9)
    z_ref.backward(g)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wn = torch.nn.Parameter(w.detach().to(DEV).contiguous(memory_format=torch.channels_last))
    rn = r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
    with torch.no_grad():
        z_single = HF.conv_bn_act(xn, wn, sc.to(DEV), sh.to(DEV), stride=stride, pad=pad, relu=True, residual=rn)
    z = HF.conv_bn_act(xn, wn, sc.to(DEV), sh.to(DEV), stride=stride, pad=pad, relu=True, residual=rn)
    assert ops.pick_split_k(N * OH * OW, O, k * k * C) >= 4          # the case does take the k-split
    close(z.permute(0, 3, 1, 2), z_ref, name="conv node fwd (k-split)")
    close(z, z_single, tol=2e-5, name="k-split vs single pass")
    z.backward(g.permute(0, 2, 3, 1).contiguous().to(DEV))
    close(xn.grad.permute(0, 3, 1, 2), x.grad, tol=2e-4, name="conv node dgrad")
    close(wn.grad, w.grad, tol=2e-4, name="conv node wgrad")


@pytest.mark.parametrize("N,H,W,C,O,k,pad,acc", [(2, 16, 16, 64, 64, 3, 1, False), (3, 8, 12, 128, 96, 3, 1, True), (2, 16, 16, 64, 128, 1, 0, False),
                                                 (1, 8, 8, 512, 512, 3, 1, True), (2, 6, 10, 32, 64, 1, 0, True)])
def test_conv_stride2_dgrad_by_parity_classes(N, H, W, C, O, k, pad, acc):
    """Data gradient of a stride-2 convolution as four stride-1 data gradients over tap sub-lattices on the half-resolution grid
    (hip/functional._dgrad_stride2, cape_gemm_desc sub-lattice fields, cape_interleave2x2_f32) against torch and against the
    one-launch gather form; with an existing gradient to accumulate onto (the bottleneck's shortcut branch)."""
    from cape_amd.hip import functional as HF
    x, w, geom, OH, OW = _conv_case(N, H, W, C, O, k, 2, pad)
    x.requires_grad_(True)
    g = rnd(N, O, OH, OW, seed=9)
    F.conv2d(x, w, stride=2, padding=pad).backward(g)
    wn = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    gn = g.permute(0, 2, 3, 1).contiguous().to(DEV)
    assert HF._dgrad_stride2_ok(geom)
    base = rnd(N, H, W, C, seed=11).to(DEV) if acc else None
    dx = base.clone() if acc else torch.empty(N, H, W, C, device=DEV)
    HF._dgrad_stride2(gn, wn, geom, dx, dx if acc else None)
    ref = x.grad.permute(0, 2, 3, 1) + (base.cpu() if acc else 0.0)
    close(dx, ref, tol=2e-4, name="stride-2 dgrad by classes")
    one = torch.empty(N * H * W, C, device=DEV)
    ops.gemm(gn, wn, one, N * H * W, C, k * k * O, a_mode=3, b_mode=2, conv=geom)
    close(dx - (base if acc else 0.0), one.view(N, H, W, C), tol=2e-5, name="classes vs one gather launch")


# ---- fragment-packed weights of the register-stationary kernel (cape_pack_weights, ops.PackedWeights) ----
@pytest.mark.parametrize("M,N,K,bm", [(5000, 256, 256, 0), (700, 384, 256, 1), (6400, 1024, 256, 0), (333, 100, 128, 1), (4097, 64, 64, 0),
                                      (6400, 70, 256, 0), (400, 70, 256, 0), (129, 33, 64, 1)])
def test_gemm_rs_packed_weights(M, N, K, bm):
    """A parameter used as the B operand is packed once into MFMA-fragment order; the packed launch must reproduce the launch
    that splits the fp32 weight itself bit for bit (same bf16 planes, same MFMA order), for row slices of a parameter too
    (MultiheadAttention.in_proj), and follow in-place updates of the weight."""
    if ops.get_gemm_precision() != "bf16x3":
        pytest.skip("packed weights feed the bf16x3 register-stationary kernel")
    ops.PackedWeights.clear()
    x = rnd(M, K, seed=1).to(DEV)
    w = torch.nn.Parameter((rnd(N, K, seed=2, scale=K ** -0.5) if bm == 0 else rnd(K, N, seed=2, scale=K ** -0.5)).to(DEV))
    b = rnd(N, seed=3).to(DEV)
    plain, packed = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.PackedWeights.enabled = False
    ops.gemm(x, w.detach(), plain, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.enabled = True
    ops.gemm(x, w, packed, M, N, K, b_mode=bm, bias=b, relu=True)
    assert len(ops.PackedWeights.entries) == 1
    assert torch.equal(plain, packed)
    # in-place edit -> version bump -> re-packed at the next use
    with torch.no_grad():
        w.mul_(0.5)
    ops.gemm(x, w, packed, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.enabled = False
    ops.gemm(x, w.detach(), plain, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.enabled = True
    assert torch.equal(plain, packed)
    # raw update behind autograd's back (what the fused AdamW kernel does) + the batched re-pack
    w.data.view(-1)[::3].add_(0.25)                   # .data: no version bump
    stale = torch.empty(M, N, device=DEV)
    ops.gemm(x, w, stale, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.invalidate_and_repack()
    ops.gemm(x, w, packed, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.enabled = False
    ops.gemm(x, w.detach(), plain, M, N, K, b_mode=bm, bias=b, relu=True)
    ops.PackedWeights.enabled = True
    assert torch.equal(plain, packed) and not torch.equal(stale, packed)
    if bm == 0 and N % 64 == 0:                       # a row slice of the parameter (its own entry)
        h = N // 2
        ops.gemm(x, w[h:], packed[:, :h], M, h, K, ldc=N)
        ops.PackedWeights.enabled = False
        ops.gemm(x, w.detach()[h:], plain[:, :h], M, h, K, ldc=N)
        ops.PackedWeights.enabled = True
        assert torch.equal(plain[:, :h], packed[:, :h]) and len(ops.PackedWeights.entries) == 2
    ops.PackedWeights.clear()


def test_gemm_mask_epilogue_and_ffn_autograd():
    """dgrad with the gate epilogue (v = mask != 0 ? v * s : 0) and the fused feed-forward node against torch autograd."""
    from cape_amd.hip import functional as HF
    M, K, Hd = 333, 256, 1024
    dy, w2, hmask = rnd(M, K, seed=1), rnd(K, Hd, seed=2, scale=K ** -0.5), F.relu(rnd(M, Hd, seed=3))
    out = torch.empty(M, Hd, device=DEV)
    ops.gemm(dy.to(DEV), w2.to(DEV), out, M, Hd, K, a_mode=0, b_mode=1, mask_src=hmask.to(DEV), mask_scale=1.25)
    close(out, (dy @ w2) * (hmask != 0) * 1.25, name="gated dgrad")
    x = rnd(2, 50, 256, seed=4).requires_grad_(True)
    w1, b1 = rnd(Hd, 256, seed=5, scale=256 ** -0.5).requires_grad_(True), rnd(Hd, seed=6).requires_grad_(True)
    w2p, b2 = rnd(256, Hd, seed=7, scale=Hd ** -0.5).requires_grad_(True), rnd(256, seed=8).requires_grad_(True)
    ref = F.linear(F.relu(F.linear(x, w1, b1)), w2p, b2)
    g = rnd(2, 50, 256, seed=9)
    ref.backward(g)
    dev = [t.detach().to(DEV).requires_grad_(True) for t in (x, w1, b1, w2p, b2)]
    y = HF.ffn(*dev)
    close(y, ref, name="ffn fwd")
    y.backward(g.to(DEV))
    for got, want, nm in zip(dev, (x, w1, b1, w2p, b2), ("dx", "dw1", "db1", "dw2", "db2")):
        close(got.grad, want.grad, tol=2e-4, name="ffn " + nm)


def test_gemm_dropout_epilogue_statistics_and_replay():
    M, N, K = 512, 256, 64
    x, w = torch.ones(M, K, device=DEV), torch.ones(N, K, device=DEV)
    rng = ops.RngState(1234, DEV)
    a, b = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm(x, w, a, M, N, K, dropout_p=0.1, rng=rng, rng_stream=5)
    ops.gemm(x, w, b, M, N, K, dropout_p=0.1, rng=rng, rng_stream=5)
    assert torch.equal(a, b)                                   # same (seed, step, stream) -> same mask
    frac = (a == 0).float().mean().item()
    assert abs(frac - 0.1) < 0.01
    assert torch.allclose(a[a != 0], torch.full_like(a[a != 0], K / 0.9))
    rng.advance()
    ops.gemm(x, w, b, M, N, K, dropout_p=0.1, rng=rng, rng_stream=5)
    assert not torch.equal(a, b)


def test_colsum():
    x = rnd(1000, 300, seed=3)
    out = torch.ones(300, device=DEV)
    ops.colsum(x.to(DEV), 1000, 300, out, accumulate=True)
    close(out, 1 + x.sum(0), tol=2e-4)


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,C", [(37, 256), (1000, 256), (9, 1024), (5, 64)])
def test_add_layernorm_fwd_bwd(rows, C):
    x, y, g, b, pos = rnd(rows, C, seed=1), rnd(rows, C, seed=2), rnd(C, seed=3) + 1, rnd(C, seed=4), rnd(rows, C, seed=5)
    x.requires_grad_(True); y.requires_grad_(True); g.requires_grad_(True); b.requires_grad_(True)
    ref = F.layer_norm(x + y, (C,), g, b, 1e-5)
    go, gp = rnd(rows, C, seed=6), rnd(rows, C, seed=7)
    (ref * go + (ref + pos) * gp).sum().backward()
    xd, yd, gd, bd, pd = (t.detach().to(DEV) for t in (x, y, g, b, pos))
    out, mean, rstd, out_pos = ops.add_layernorm_fwd(xd, yd, gd, bd, pos=pd)
    close(out, ref, name="ln out")
    close(out_pos, ref + pos, name="ln out_pos")
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx, dy = ops.add_layernorm_bwd(go.to(DEV), gp.to(DEV), xd, yd, gd, mean, rstd, dg, db)
    close(dx, x.grad, tol=2e-4, name="ln dx")
    close(dy, y.grad, tol=2e-4, name="ln dy")
    close(dg, g.grad, tol=2e-4, name="ln dgamma")
    close(db, b.grad, tol=2e-4, name="ln dbeta")


def test_add_layernorm_dropout_consistency():
    rows, C = 64, 256
    x, y = torch.zeros(rows, C, device=DEV), torch.ones(rows, C, device=DEV)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    rng = ops.RngState(7, DEV)
    out, mean, rstd, _ = ops.add_layernorm_fwd(x, y, g, b, dropout_p=0.25, rng=rng, rng_stream=3)
    kept = out > 0                                             # s in {0, 1/0.75}: kept elements are above the row mean
    frac = 1 - kept.float().mean().item()
    assert abs(frac - 0.25) < 0.03
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx, dy = ops.add_layernorm_bwd(torch.randn(rows, C, device=DEV), None, x, y, g, mean, rstd, dg, db, dropout_p=0.25,
                                   rng=rng, rng_stream=3)
    assert torch.equal(dy == 0, ~kept) or ((dy == 0) & kept).float().mean() < 1e-3
    assert torch.allclose(dy[kept], dx[kept] / 0.75)


@pytest.mark.parametrize("N,HW,C", [(3, 64, 256), (2, 1024, 256), (2, 1, 256)])
def test_groupnorm_fwd_bwd(N, HW, C):
    x = rnd(N, C, HW, seed=1)
    g, b = rnd(C, seed=2) + 1, rnd(C, seed=3)
    x.requires_grad_(True); g.requires_grad_(True); b.requires_grad_(True)
    ref = F.group_norm(x, 32, g, b, 1e-5)
    go = rnd(N, C, HW, seed=4)
    (ref * go).sum().backward()
    xd = x.detach().permute(0, 2, 1).contiguous().to(DEV)                    # (N,HW,C)
    S = HW + 5
    out = torch.zeros(N, S, C, device=DEV)
    mean, rstd = ops.groupnorm_fwd(xd, g.detach().to(DEV), b.detach().to(DEV), out[:, 3:], S * C, N, HW, C)
    close(out[:, 3:3 + HW], ref.permute(0, 2, 1), name="gn out")
    assert float(out[:, :3].abs().sum()) == 0 and float(out[:, 3 + HW:].abs().sum()) == 0
    dfull = torch.zeros(N, S, C, device=DEV)
    dfull[:, 3:3 + HW] = go.permute(0, 2, 1).to(DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.groupnorm_bwd(dfull[:, 3:], S * C, xd, g.detach().to(DEV), mean, rstd, dg, db, N, HW, C)
    close(dx, x.grad.permute(0, 2, 1), tol=2e-4, name="gn dx")
    close(dg, g.grad, tol=2e-4, name="gn dgamma")
    close(db, b.grad, tol=2e-4, name="gn dbeta")


# ------------------------------------------------------------------------------------------------
# MSDA
# ------------------------------------------------------------------------------------------------
def _msda_inputs(N, Lq, shapes, seed):
    S = sum(h * w for h, w in shapes)
    value = rnd(N, S, 8, 32, seed=seed)
    offw = torch.cat([rnd(N, Lq, 256, seed=seed + 1, scale=1.5), rnd(N, Lq, 128, seed=seed + 2)], -1).contiguous()
    ref = torch.rand(N, Lq, len(shapes), 2, generator=torch.Generator().manual_seed(seed + 3)) * 1.2 - 0.1
    return value, offw, ref


def _msda_ref(value, offw, ref, shapes):
    N, Lq = offw.shape[:2]
    L = len(shapes)
    off = offw[..., :256].reshape(N, Lq, 8, L, 4, 2)
    aw = F.softmax(offw[..., 256:].reshape(N, Lq, 8, 16), -1).view(N, Lq, 8, L, 4)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    return cape_ref.msda_core(value, shapes, loc, aw)


@pytest.mark.parametrize("N,Lq,shapes", [(2, 37, [(8, 8), (4, 4), (2, 2), (1, 1)]), (9, 5, [(6, 10), (3, 5), (2, 3), (1, 2)]),
                                         (1, 1360, [(32, 32), (16, 16), (8, 8), (4, 4)]),
                                         (3, 70, [(32, 32), (16, 16), (8, 8), (4, 4)]),
                                         (2, 90, [(48, 48), (24, 24), (12, 12), (6, 6)]),      # S = 3060: 4-channel slabs
                                         (1, 130, [(64, 64), (32, 32), (16, 16), (8, 8)])])    # S = 5440: 2-channel slabs
@pytest.mark.parametrize("form", ["f64", "fx", "atomic"])          # d_value accumulators: fp64 LDS slab | fixed-point pairs | memory atomics
def test_msda_fwd_bwd(N, Lq, shapes, form):
    value, offw, ref = _msda_inputs(N, Lq, shapes, 11)
    value.requires_grad_(True); offw.requires_grad_(True); ref.requires_grad_(True)
    out_ref = _msda_ref(value, offw, ref, shapes)
    go = rnd(N, Lq, 256, seed=5)
    out_ref.backward(go)
    geo = ops.LevelGeometry(shapes)
    vd, od, rd = value.detach().to(DEV), offw.detach().to(DEV), ref.detach().to(DEV)
    out = ops.msda_fwd(vd, od, rd, geo, N, Lq)
    close(out, out_ref, name="msda fwd")
    dv, do, dr = ops.msda_bwd(go.to(DEV), vd, od, rd, geo, N, Lq, form=form)
    close(dv, value.grad, tol=3e-4, name="msda d_value")
    close(do, offw.grad, tol=3e-4, name="msda d_offw")
    close(dr, ref.grad, tol=3e-4, name="msda d_ref")
    dv2, do2, dr2 = ops.msda_bwd(go.to(DEV), vd, od, rd, geo, N, Lq, need_ref_grad=False, form=form)
    assert dr2 is None
    close(dv2, value.grad, tol=3e-4, name="msda d_value (no d_ref)")
    close(do2, offw.grad, tol=3e-4, name="msda d_offw (no d_ref)")


def test_msda_value_gradient_fixed_point_bounds():
    """The fixed-point d_value slab (csrc/msda.hip msda_bwd_value_fx_kernel) against the fp64 slab on the cases its scale is
    chosen for: (a) every query of an image puts all its weight on ONE pixel (the largest sum the softmax allows: Lq * gmax --
    must not overflow); (b) channels 1e6 apart in one block (the small channel is quantised against the block's largest
    gradient: absolute, not relative, accuracy); (c) an all-zero and a denormal-small d_out; (d) a NaN in d_out poisons the
    (image, channel block) slice it belongs to instead of vanishing in the integer conversion."""
    shapes = [(32, 32), (16, 16), (8, 8), (4, 4)]
    geo = ops.LevelGeometry(shapes)
    N, Lq = 2, 1360
    value, offw, ref = _msda_inputs(N, Lq, shapes, 21)
    vd = value.to(DEV)

    def both(go, offw_, ref_):
        a = ops.msda_bwd(go.to(DEV), vd, offw_.to(DEV), ref_.to(DEV), geo, N, Lq, form="f64")[0]
        b = ops.msda_bwd(go.to(DEV), vd, offw_.to(DEV), ref_.to(DEV), geo, N, Lq, form="fx")[0]
        return a.cpu(), b.cpu()

    def quantum(gmax):                                           # pow2ceil(Lq) * pow2ceil(gmax) * 2^-30
        return 2.0 ** (11 + int(np.ceil(np.log2(gmax))) - 30)

    # (a) collision: zero offsets, one reference point at a pixel centre of level 0, logits select sample 0 only
    offw_c = torch.zeros(N, Lq, 384)
    offw_c[..., 256:] = -1e4
    offw_c[..., 256::16] = 0.0                                   # sample 0 of every head
    ref_c = torch.full((N, Lq, 4, 2), (5 + 0.5) / 32)
    go = torch.ones(N, Lq, 256) * 3.0
    a, b = both(go, offw_c, ref_c)
    assert abs(float(a.abs().max()) - 3.0 * Lq) < 1e-2           # the whole image's gradient on one pixel
    assert (a - b).abs().max() <= Lq * 0.5 * quantum(3.0) + 1e-3
    # (b) dynamic range inside a block + generic taps
    go = rnd(N, Lq, 256, seed=6)
    go[..., 1::16] *= 1e-6
    a, b = both(go, offw, ref)
    gmax = float(go.abs().max())
    assert (a - b).abs().max() <= 64 * quantum(gmax), ((a - b).abs().max(), quantum(gmax))
    small = (a.view(N, -1, 256)[..., 1::16] - b.view(N, -1, 256)[..., 1::16]).abs().max()
    assert small <= 64 * quantum(gmax)                           # absolute bound only: the small channel's own scale is 1e-6 gmax
    # (c) zero and tiny gradients
    a, b = both(torch.zeros(N, Lq, 256), offw, ref)
    assert float(b.abs().max()) == 0.0 and float(a.abs().max()) == 0.0
    a, b = both(rnd(N, Lq, 256, seed=7) * 1e-30, offw, ref)
    assert torch.isfinite(b).all() and (a - b).abs().max() <= 64 * quantum(4e-30)
    # (d) NaN
    go = rnd(N, Lq, 256, seed=8)
    go[1, 77, 40] = float("nan")                                 # image 1, head 1; the block is 4..16 channels wide (by image count)
    a, b = both(go, offw, ref)
    bv = b.view(N, -1, 256)
    assert torch.isnan(bv[1, :, 40:44]).all() and torch.isfinite(bv[0]).all()
    assert torch.isfinite(bv[1, :, :32]).all() and torch.isfinite(bv[1, :, 48:]).all()


def test_msda_core_against_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "msda_core.npz"))
    shapes = [tuple(int(v) for v in s) for s in d["shapes"]]
    value, loc, aw = torch.from_numpy(d["value"]), torch.from_numpy(d["loc"]), torch.from_numpy(d["aw"])
    N, Lq = loc.shape[:2]
    # express (loc, softmaxed aw) through the kernel's inputs: ref = 0, offsets = loc * (W,H), logits = log(aw)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32)
    off = (loc * norm[None, None, None, :, None, :]).reshape(N, Lq, 256)
    offw = torch.cat([off, aw.clamp_min(1e-30).log().reshape(N, Lq, 128)], -1).contiguous()
    ref = torch.zeros(N, Lq, 4, 2)
    out = ops.msda_fwd(value.to(DEV), offw.to(DEV), ref.to(DEV), ops.LevelGeometry(shapes), N, Lq)
    close(out, torch.from_numpy(d["out"]), name="msda golden")


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale, mask):
    N, Lq, C = q.shape
    H = C // 32
    qh, kh, vh = (t.view(N, -1, H, 32).transpose(1, 2) for t in (q, k, v))
    s = (qh * scale) @ kh.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    return (F.softmax(s, -1) @ vh).transpose(1, 2).reshape(N, Lq, C)


@pytest.mark.parametrize("N,Lq,Lk,mode", [(2, 200, 200, 1), (3, 70, 17, 2), (2, 9, 9, 0), (2, 1, 37, 0), (1, 130, 130, 1),
                                          (2, 24, 280, 0), (1, 300, 290, 1), (2, 280, 24, 2)])        # panels beyond 256 rows
def test_attention_fwd_bwd(N, Lq, Lk, mode):
    H = 8
    big_q, big_k, big_v = rnd(N, Lq, 768, seed=1), rnd(N, Lk, 768, seed=2), rnd(N, Lk, 768, seed=3)
    q, k, v = big_q[..., :256], big_k[..., 256:512], big_v[..., 512:]          # strided views (ld = 768)
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    mask, kpm = None, None
    if mode == 1:
        mask = torch.triu(torch.full((Lq, Lk), float("-inf")), diagonal=1)
    if mode == 2:
        kpm = torch.zeros(N, Lk, dtype=torch.bool)
        kpm[:, 3] = True; kpm[0, 10:] = True
        mask = torch.zeros(N, 1, 1, Lk).masked_fill(kpm[:, None, None, :], float("-inf"))
    scale = 32 ** -0.5
    ref = _attn_ref(qc, kc, vc, scale, mask)
    go = rnd(N, Lq, 256, seed=4)
    ref.backward(go)
    Qd, Kd, Vd = big_q.to(DEV), big_k.to(DEV), big_v.to(DEV)
    qd, kd, vd = Qd[..., :256], Kd[..., 256:512], Vd[..., 512:]
    kpm_d = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    O, lse = ops.attn_fwd(qd, kd, vd, N, H, Lq, Lk, scale, mask_mode=mode, kpm=kpm_d)
    close(O, ref, name="attn fwd")
    dQ, dK, dV = torch.zeros_like(Qd), torch.zeros_like(Kd), torch.zeros_like(Vd)
    ops.attn_bwd(go.to(DEV), qd, kd, vd, O, lse, dQ[..., :256], dK[..., 256:512], dV[..., 512:], N, H, Lq, Lk, scale,
                 mask_mode=mode, kpm=kpm_d)
    close(dQ[..., :256], qc.grad, tol=2e-4, name="attn dQ")
    close(dK[..., 256:512], kc.grad, tol=2e-4, name="attn dK")
    close(dV[..., 512:], vc.grad, tol=2e-4, name="attn dV")
    assert float(dQ[..., 256:].abs().sum()) == 0


@pytest.mark.parametrize("N,Lq,Lk,mode", [(2, 300, 700, 0), (1, 1360, 40, 0), (2, 70, 513, 2), (1, 600, 600, 1)])
def test_attention_fwd_long_sequences(N, Lq, Lk, mode):
    """Forward with more than 256 keys / queries (keys tiled 256 rows at a time): the bidirectional attention blocks."""
    H = 8
    q, k, v = rnd(N, Lq, 256, seed=1), rnd(N, Lk, 256, seed=2), rnd(N, Lk, 256, seed=3)
    mask, kpm = None, None
    if mode == 1:
        mask = torch.triu(torch.full((Lq, Lk), float("-inf")), diagonal=1)
    if mode == 2:
        kpm = torch.zeros(N, Lk, dtype=torch.bool)
        kpm[:, 3] = True; kpm[0, 300:] = True
        mask = torch.zeros(N, 1, 1, Lk).masked_fill(kpm[:, None, None, :], float("-inf"))
    ref = _attn_ref(q, k, v, 32 ** -0.5, mask)
    kpm_d = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    O, lse = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), N, H, Lq, Lk, 32 ** -0.5, mask_mode=mode, kpm=kpm_d)
    close(O, ref, name="attn fwd long")


@pytest.mark.parametrize("N,Lk,mode", [(32, 200, 0), (3, 17, 2), (2, 1, 0), (2, 300, 2), (4, 77, 1)])
def test_attention_single_query_decode_form(N, Lk, mode):
    """Lq == 1 takes the one-wave-per-(image, head) kernel: strided K/V caches, key padding, causal offset."""
    H, cap = 8, Lk + 5
    q = rnd(N, 1, 256, seed=1)
    kc, vc = rnd(N, cap, 256, seed=2), rnd(N, cap, 256, seed=3)            # caches longer than the filled part
    k, v = kc[:, :Lk], vc[:, :Lk]
    mask, kpm, off, Lk_eff = None, None, 0, Lk
    if mode == 2:
        kpm = torch.zeros(N, Lk, dtype=torch.bool)
        kpm[:, Lk // 2] = True; kpm[0, Lk - 1] = True
        if Lk > 1:
            mask = torch.zeros(N, 1, 1, Lk).masked_fill(kpm[:, None, None, :], float("-inf"))
        else:
            kpm = None
    if mode == 1:
        off = Lk // 2                                                      # query sees keys 0..off
        mask = torch.zeros(1, Lk).masked_fill(torch.arange(Lk)[None] > off, float("-inf"))
    ref = _attn_ref(q, k, v, 32 ** -0.5, mask)
    kpm_d = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    O, lse = ops.attn_fwd(q.to(DEV), kc.to(DEV)[:, :Lk], vc.to(DEV)[:, :Lk], N, H, 1, Lk, 32 ** -0.5,
                          mask_mode=(2 if kpm is not None else (1 if mode == 1 else 0)), causal_offset=off, kpm=kpm_d)
    close(O, ref, name="attn decode form")


def test_gelu_and_scale_residual():
    x, y, g = rnd(37, 256, seed=1, scale=2.0), rnd(37, 256, seed=2), rnd(256, seed=3)
    close(ops.gelu(x.to(DEV)), F.gelu(x), tol=1e-6, name="gelu")
    close(ops.scale_residual(x.to(DEV), y.to(DEV), g.to(DEV)), x + y * g, tol=1e-6, name="scale_residual")
    close(ops.scale_residual(x.to(DEV), y.to(DEV)), x + y, tol=1e-6, name="residual")


@pytest.mark.parametrize("N,Lq,Lk,mode", [(2, 200, 200, 1), (3, 72, 128, 2), (1, 64, 260, 0)])
@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_attention_matrix_core_form(N, Lq, Lk, mode, prec):
    """Batched-GEMM attention (Q K^T, P V, dV, dP, dQ, dK on the matrix cores + row softmax kernels) against torch autograd,
    on strided head-interleaved views like the fused kernels."""
    H = 8
    big_q, big_k, big_v = rnd(N, Lq, 768, seed=1), rnd(N, Lk, 768, seed=2), rnd(N, Lk, 768, seed=3)
    q, k, v = big_q[..., :256], big_k[..., 256:512], big_v[..., 512:]
    qc, kc, vc = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
    mask, kpm = None, None
    if mode == 1:
        mask = torch.triu(torch.full((Lq, Lk), float("-inf")), diagonal=1)
    if mode == 2:
        kpm = torch.zeros(N, Lk, dtype=torch.bool)
        kpm[:, 3] = True; kpm[0, 100:] = True
        mask = torch.zeros(N, 1, 1, Lk).masked_fill(kpm[:, None, None, :], float("-inf"))
    scale = 32 ** -0.5
    ref = _attn_ref(qc, kc, vc, scale, mask)
    go = rnd(N, Lq, 256, seed=4)
    ref.backward(go)
    Qd, Kd, Vd = big_q.to(DEV), big_k.to(DEV), big_v.to(DEV)
    qd, kd, vd = Qd[..., :256], Kd[..., 256:512], Vd[..., 512:]
    kpm_d = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    old = ops.get_gemm_precision()
    try:
        ops.set_gemm_precision(prec)
        assert ops.attn_mm_ok(N, H, Lq, Lk)
        O, P, Pu = ops.attn_mm_fwd(qd, kd, vd, N, H, Lq, Lk, scale, mask_mode=mode, kpm=kpm_d)
        dQ, dK, dV = torch.zeros_like(Qd), torch.zeros_like(Kd), torch.zeros_like(Vd)
        ops.attn_mm_bwd(go.to(DEV), qd, kd, vd, P, Pu, dQ[..., :256], dK[..., 256:512], dV[..., 512:], N, H, Lq, Lk, scale)
    finally:
        ops.set_gemm_precision(old)
    tol = 1e-4 if prec == "f32" else 3e-4
    close(O, ref, tol=tol, name="attn mm fwd")
    close(dQ[..., :256], qc.grad, tol=2 * tol, name="attn mm dQ")
    close(dK[..., 256:512], kc.grad, tol=2 * tol, name="attn mm dK")
    close(dV[..., 512:], vc.grad, tol=2 * tol, name="attn mm dV")
    assert float(dQ[..., 256:].abs().sum()) == 0
    # same probabilities as the fused kernel (same masks)
    O2, _ = ops.attn_fwd(qd, kd, vd, N, H, Lq, Lk, scale, mask_mode=mode, kpm=kpm_d)
    close(O, O2.cpu(), tol=tol, name="attn mm vs fused")


def test_attention_matrix_core_form_dropout_consistent():
    """Dropout: same keep decisions as the fused kernel (same counter index), and the gradient of its own forward."""
    N, H, L = 2, 8, 128
    q, k, v = rnd(N, L, 256, seed=1).to(DEV), rnd(N, L, 256, seed=2).to(DEV), rnd(N, L, 256, seed=3).to(DEV)
    g = rnd(N, L, 256, seed=4).to(DEV)
    rng = ops.RngState(99, DEV)
    O, P, Pu = ops.attn_mm_fwd(q, k, v, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    Of, _ = ops.attn_fwd(q, k, v, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    close(O, Of.cpu(), tol=3e-4, name="attn mm dropout vs fused")
    dQ, dK, dV = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_mm_bwd(g, q, k, v, P, Pu, dQ, dK, dV, N, H, L, L, 0.2, dropout_p=0.3, rng=rng, rng_stream=9)
    dQf, dKf, dVf = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_bwd(g, q, k, v, Of, _, dQf, dKf, dVf, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    close(dQ, dQf.cpu(), tol=5e-4, name="dQ"); close(dK, dKf.cpu(), tol=5e-4, name="dK"); close(dV, dVf.cpu(), tol=5e-4, name="dV")


@pytest.mark.parametrize("N,Lq,Lk,mode", [(2, 200, 200, 1), (3, 72, 128, 2), (1, 64, 200, 0), (2, 224, 224, 1), (1, 33, 97, 1), (2, 200, 40, 0),
                                          (3, 200, 17, 2), (4, 17, 17, 2), (2, 9, 9, 0), (2, 1, 37, 0)])      # few keys / few queries
def test_flash_attention_fwd_bwd(N, Lq, Lk, mode):
    """csrc/flash_attn.hip (fused Q K^T -> mask -> softmax -> P V on the matrix cores, backward recomputed from the log-sum-exp)
    against the plain torch expression on the CPU: strided q | k | v views of one (N, L, 768) buffer as the decoder passes
    them, every mask mode, ragged last blocks."""
    H = 8
    big_q, big_k, big_v = rnd(N, Lq, 768, seed=1), rnd(N, Lk, 768, seed=2), rnd(N, Lk, 768, seed=3)
    q, k, v = big_q[..., :256], big_k[..., 256:512], big_v[..., 512:]
    qc, kc, vc = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
    mask, kpm = None, None
    if mode == 1:
        mask = torch.triu(torch.full((Lq, Lk), float("-inf")), diagonal=1)
    if mode == 2:
        kpm = torch.zeros(N, Lk, dtype=torch.bool)
        kpm[:, 3] = True; kpm[0, 100:] = True
        mask = torch.zeros(N, 1, 1, Lk).masked_fill(kpm[:, None, None, :], float("-inf"))
    scale = 32 ** -0.5
    ref = _attn_ref(qc, kc, vc, scale, mask)
    go = rnd(N, Lq, 256, seed=4)
    ref.backward(go)
    Qd, Kd, Vd = big_q.to(DEV), big_k.to(DEV), big_v.to(DEV)
    qd, kd, vd = Qd[..., :256], Kd[..., 256:512], Vd[..., 512:]
    kpm_d = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    O, lse = ops.flash_attn_fwd(qd, kd, vd, N, H, Lq, Lk, scale, mask_mode=mode, kpm=kpm_d)
    close(O, ref, tol=1e-4, name="flash fwd")
    sref = (qc.detach().view(N, Lq, H, 32).transpose(1, 2) * scale) @ kc.detach().view(N, Lk, H, 32).transpose(1, 2).transpose(-1, -2)
    if mask is not None:
        sref = sref + mask
    close(lse, torch.logsumexp(sref, -1), tol=1e-4, name="flash lse")
    dQ, dK, dV = torch.zeros_like(Qd), torch.zeros_like(Kd), torch.zeros_like(Vd)
    ops.flash_attn_bwd(go.to(DEV), qd, kd, vd, O, lse, dQ[..., :256], dK[..., 256:512], dV[..., 512:], N, H, Lq, Lk, scale,
                       mask_mode=mode, kpm=kpm_d)
    close(dQ[..., :256], qc.grad, tol=2e-4, name="flash dQ")
    close(dK[..., 256:512], kc.grad, tol=2e-4, name="flash dK")
    close(dV[..., 512:], vc.grad, tol=2e-4, name="flash dV")
    assert float(dQ[..., 256:].abs().sum()) == 0 and float(dK[..., :256].abs().sum()) == 0


def test_flash_attention_dropout_matches_the_batched_gemm_form():
    """Same dropout element index in every attention form: with the same (seed, step, stream) the fused kernel drops exactly the
    probabilities the batched-GEMM form drops -- outputs and all three gradients agree."""
    N, H, L = 2, 8, 200
    q, k, v = rnd(N, L, 256, seed=1).to(DEV), rnd(N, L, 256, seed=2).to(DEV), rnd(N, L, 256, seed=3).to(DEV)
    g = rnd(N, L, 256, seed=4).to(DEV)
    rng = ops.RngState(99, DEV)
    O, P, Pu = ops.attn_mm_fwd(q, k, v, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    Of, lse = ops.flash_attn_fwd(q, k, v, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    close(Of, O.cpu(), tol=3e-4, name="flash dropout vs batched form")
    dQ, dK, dV = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_mm_bwd(g, q, k, v, P, Pu, dQ, dK, dV, N, H, L, L, 0.2, dropout_p=0.3, rng=rng, rng_stream=9)
    dQf, dKf, dVf = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.flash_attn_bwd(g, q, k, v, Of, lse, dQf, dKf, dVf, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    close(dQf, dQ.cpu(), tol=5e-4, name="dQ"); close(dKf, dK.cpu(), tol=5e-4, name="dK"); close(dVf, dV.cpu(), tol=5e-4, name="dV")


def test_attention_dropout_fwd_bwd_consistent():
    """With dropout the kernel's gradients must be the gradients of its own (masked) forward:
    finite-difference check of sum(O * G) with respect to V (linear in V -> exact up to rounding)."""
    N, H, L = 1, 8, 40
    q, k, v = rnd(N, L, 256, seed=1).to(DEV), rnd(N, L, 256, seed=2).to(DEV), rnd(N, L, 256, seed=3).to(DEV)
    g = rnd(N, L, 256, seed=4).to(DEV)
    rng = ops.RngState(99, DEV)
    O, lse = ops.attn_fwd(q, k, v, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    dQ, dK, dV = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_bwd(g, q, k, v, O, lse, dQ, dK, dV, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    dv_dir = rnd(N, L, 256, seed=5).to(DEV)
    O2, _ = ops.attn_fwd(q, k, v + dv_dir, N, H, L, L, 0.2, mask_mode=1, dropout_p=0.3, rng=rng, rng_stream=9)
    lhs = ((O2 - O) * g).sum().item()
    rhs = (dV * dv_dir).sum().item()
    assert abs(lhs - rhs) <= 2e-3 * max(1.0, abs(rhs))


# ------------------------------------------------------------------------------------------------
# elementwise / small ops
# ------------------------------------------------------------------------------------------------
def test_layout_bn_maxpool_relu_bwd():
    x = rnd(2, 3, 9, 7, seed=1)
    out = ops.nchw_to_nhwc(x.to(DEV), 4)
    close(out[..., :3], x.permute(0, 2, 3, 1))
    assert float(out[..., 3].abs().sum()) == 0
    w, b, rm, rv = rnd(64, seed=2) + 1, rnd(64, seed=3), rnd(64, seed=4), rnd(64, seed=5).abs() + 0.5
    sc, sh = ops.bn_fold(w.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV))
    s_ref = w * (rv + 1e-5).rsqrt()
    close(sc, s_ref, tol=1e-6); close(sh, b - rm * s_ref, tol=1e-6)
    y = rnd(2, 64, 11, 13, seed=6)
    mp = ops.maxpool3x3s2(y.permute(0, 2, 3, 1).contiguous().to(DEV))
    close(mp.permute(0, 3, 1, 2), F.max_pool2d(y, 3, 2, 1), tol=0)
    dy, yy = rnd(50, 64, seed=7), rnd(50, 64, seed=8)
    d_pre, d_res = ops.bn_relu_bwd(dy.to(DEV), yy.to(DEV), sc, True, True)
    close(d_res, dy * (yy > 0)); close(d_pre, dy * (yy > 0) * s_ref)
    close(ops.relu_drop_bwd(dy.to(DEV), yy.to(DEV), 1 / 0.9), dy * (yy > 0) / 0.9)
    close(ops.add(dy.to(DEV), yy.to(DEV)), dy + yy, tol=0)


def test_pos_sine_level():
    N, h, w = 2, 6, 5
    mask = torch.zeros(N, h, w, dtype=torch.bool)
    mask[1, 4:, :] = True; mask[1, :, 3:] = True
    lvl = rnd(256, seed=1)
    S = h * w + 4
    out = torch.zeros(N, S, 256, device=DEV)
    ops.pos_sine_level(mask.to(torch.uint8).to(DEV), lvl.to(DEV), out[:, 2:], S * 256, N, h, w)
    ref = cape_ref.position_embedding_sine(mask).reshape(N, h * w, 256) + lvl
    close(out[:, 2:2 + h * w], ref, tol=2e-6)


def test_token_embed_query_sine_refine():
    R, V, C = 300, 2000, 256
    tab = rnd(V, C, seed=1); tab[1939] = 0
    g = torch.Generator().manual_seed(2)
    seqs = [torch.randint(0, 1940, (R,), generator=g) for _ in range(4)]
    seqs[0][:5] = 1939
    deltas = [torch.rand(R, generator=g) for _ in range(4)]
    t = {"seq11": seqs[0], "seq21": seqs[1], "seq12": seqs[2], "seq22": seqs[3],
         "delta_x1": deltas[0], "delta_x2": deltas[1], "delta_y1": deltas[2], "delta_y2": deltas[3]}
    tabg = tab.clone().requires_grad_(True)
    ref = cape_ref.seq_embed({"base_model.transformer.decoder.token_embed.weight": tabg}, t)
    go = rnd(R, C, seed=3)
    ref.backward(go)
    sd, dd = [s.to(DEV) for s in seqs], [d.to(DEV) for d in deltas]
    out = ops.token_embed_fwd(tab.to(DEV), sd, dd)
    close(out, ref, tol=1e-6, name="tok fwd")
    dt = torch.zeros(V, C, device=DEV)
    ops.token_embed_bwd(go.to(DEV), sd, dd, dt, 1939)
    want = tabg.grad.clone(); want[1939] = 0                    # padding_idx row receives no gradient
    close(dt, want, tol=1e-4, name="tok bwd")

    refp = torch.rand(R, 2, generator=g).requires_grad_(True)
    qs = cape_ref.query_pos_sine(refp[None])[0]
    gq = rnd(R, 256, seed=4)
    qs.backward(gq)
    close(ops.query_sine_fwd(refp.detach().to(DEV)), qs, tol=2e-6, name="qsine fwd")
    close(ops.query_sine_bwd(gq.to(DEV), refp.detach().to(DEV)), refp.grad, tol=2e-4, name="qsine bwd")

    r0 = torch.rand(R, 2, generator=g); r0[0, 0] = 0.0; r0[1, 1] = 1.0; r0[2, 0] = 1e-6
    r0.requires_grad_(True)
    dl = rnd(R, 2, seed=5).requires_grad_(True)
    nr = torch.sigmoid(dl + cape_ref.inverse_sigmoid(r0))
    gn = rnd(R, 2, seed=6)
    nr.backward(gn)
    nd = ops.refine_fwd(dl.detach().to(DEV), r0.detach().to(DEV))
    close(nd, nr, tol=2e-6, name="refine fwd")
    dd_, dr_ = ops.refine_bwd(gn.to(DEV), nd, r0.detach().to(DEV))
    close(dd_, dl.grad, tol=1e-5, name="refine d_delta")
    close(dr_[3:], r0.grad[3:], tol=2e-4, name="refine d_ref")
    x = rnd(77, seed=7)
    y = ops.sigmoid_fwd(x.to(DEV))
    close(y, torch.sigmoid(x), tol=1e-6)
    close(ops.sigmoid_bwd(torch.ones(77, device=DEV), y), torch.sigmoid(x) * (1 - torch.sigmoid(x)), tol=1e-6)
    vr = torch.rand(3, 4, 2, generator=g)
    rr = torch.rand(3 * 10, 2, generator=g)
    o = ops.ref_scale_fwd(rr.to(DEV), vr.to(DEV), 10, 4)
    want = rr.view(3, 10, 1, 2) * vr[:, None]
    close(o, want.reshape(30, 4, 2), tol=1e-6)
    gi = rnd(30, 4, 2, seed=8)
    close(ops.ref_scale_bwd(gi.to(DEV), vr.to(DEV), 10, 4), (gi.view(3, 10, 4, 2) * vr[:, None]).sum(2).reshape(30, 2), tol=1e-5)


# ------------------------------------------------------------------------------------------------
# support encoder pieces
# ------------------------------------------------------------------------------------------------
def test_support_embed_adjacency_gcn(golden_dir, proc_sd):
    d = np.load(os.path.join(golden_dir, "support_encoder.npz"))
    coords, m = torch.from_numpy(d["coords"]), torch.from_numpy(d["enc_mask"])
    skel = json.loads(bytes(d["skel_json"]).decode())
    N, P = coords.shape[:2]
    sd = proc_sd
    W0, b0 = sd["support_encoder.coord_mlp.0.weight"], sd["support_encoder.coord_mlp.0.bias"]
    pe1d = sd["support_encoder.sequence_pos_encoding.pe"][0]
    h, pe = ops.support_embed_fwd(coords.to(DEV), W0.to(DEV), b0.to(DEV), pe1d.contiguous().to(DEV), N, P)
    close(h, F.relu(F.linear(coords, W0, b0)).reshape(N * P, 256), tol=1e-6, name="coord mlp0")
    close(pe, (cape_ref.support_pe_2d(coords) + pe1d[None, :P]).reshape(N * P, 256), tol=2e-6, name="support pe")
    gh = rnd(N * P, 256, seed=1)
    dW, db = torch.zeros(256, 2, device=DEV), torch.zeros(256, device=DEV)
    ops.support_embed_bwd(gh.to(DEV), h, coords.to(DEV), dW, db, N, P)
    gm = gh * (h.cpu() > 0)
    close(dW, gm.t() @ coords.reshape(-1, 2), tol=1e-4); close(db, gm.sum(0), tol=1e-4)
    edges = [e for s in skel for e in s]
    start = np.cumsum([0] + [len(s) for s in skel]).astype(np.int32)
    e_t = torch.tensor(edges if edges else [[0, 0]], dtype=torch.int32).to(DEV)
    adj = ops.adjacency(e_t, torch.from_numpy(start).to(DEV), m.to(torch.uint8).to(DEV), N, P)
    close(adj, torch.from_numpy(d["adj"]), tol=1e-7, name="adjacency")
    y = rnd(N, P, 512, seed=2).requires_grad_(True)
    adj_c = torch.from_numpy(d["adj"])
    ref = F.relu(torch.einsum("nvkc,nkvw->nwc", y.view(N, P, 2, 256), adj_c))
    go = rnd(N, P, 256, seed=3)
    ref.backward(go)
    out = ops.gcn_aggregate_fwd(y.detach().to(DEV), adj, N, P)
    close(out, ref, name="gcn fwd")
    close(ops.gcn_aggregate_bwd(go.to(DEV), out, adj, N, P), y.grad, name="gcn bwd")
    x = rnd(N * P, 256, seed=4).to(DEV)
    rm = torch.zeros(N * P, dtype=torch.uint8); rm[P:2 * P] = 1
    ops.zero_rows(x, rm.to(DEV))
    assert float(x[P:2 * P].abs().sum()) == 0 and float(x[:P].abs().sum()) > 0


# ------------------------------------------------------------------------------------------------
# loss, optimizer, decode bookkeeping
# ------------------------------------------------------------------------------------------------
def test_loss_fwd_bwd():
    from oracle import synth
    b = synth.make_batch(5, 3, 2, 64, 9, CFG, n_invisible=(2, 0))
    t = b["targets"]
    N, L = t["token_labels"].shape
    logits = rnd(6, N, L, 3, seed=1).requires_grad_(True)
    coords = torch.rand(6, N, L, 2, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
    out = {"pred_logits": logits[5], "pred_coords": coords[5],
           "aux_outputs": [{"pred_logits": logits[i], "pred_coords": coords[i]} for i in range(5)]}
    losses, w, total = cape_ref.criterion(out, t, CFG)
    (total * 0.25).backward()
    cw = torch.tensor([1.0, 1.0, 20.0])
    ls, tot, dl, dc = ops.loss_fwd_bwd(logits.detach().to(DEV), coords.detach().to(DEV), t["token_labels"].to(DEV),
                                       t["visibility_mask"].to(torch.uint8).to(DEV), t["target_seq"].to(DEV), cw.to(DEV),
                                       1.0, 5.0, 0.25)
    ls = ls.cpu()
    names = [f"_{i}" for i in range(5)] + [""]
    for i, s in enumerate(names):
        assert abs(ls[2 * i].item() - float(losses["loss_ce" + s])) < 1e-5
        assert abs(ls[2 * i + 1].item() - float(losses["loss_coords" + s])) < 1e-6
    assert abs(tot.item() - float(total)) < 1e-4
    close(dl, logits.grad, tol=1e-6, name="d_logits")
    close(dc, coords.grad, tol=1e-6, name="d_coords")


def test_adamw_matches_torch():
    n = 10007
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-2)
    pd, md, vd = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    from cape_amd.hip import lib
    ss = torch.full((lib.SUMSQ_PARTS,), 7.0, device=DEV)       # partial sums: every slot is rewritten by cape_sumsq
    for it in range(3):
        g = g0 * (it + 1)
        p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 0.1)
        opt.step()
        gd = g.to(DEV)
        ops.sumsq(gd, ss)
        ops.step_increment(step)
        ops.adamw_step(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.1, ss, step)
    assert abs(math.sqrt(ss.sum().item()) - (g0 * 3).norm().item()) < 1e-2
    close(pd, p.detach(), tol=1e-6, name="adamw")
    # no atomics anywhere in the norm: two evaluations agree bit for bit (data-parallel replicas must clip identically)
    big = rnd(3_000_001, seed=9).to(DEV)
    s1, s2 = torch.empty(lib.SUMSQ_PARTS, device=DEV), torch.empty(lib.SUMSQ_PARTS, device=DEV)
    ops.sumsq(big, s1); ops.sumsq(big, s2)
    assert torch.equal(s1, s2) and abs(s1.sum().item() / float((big.double() ** 2).sum()) - 1.0) < 1e-5


def test_decode_next_tokens_matches_oracle():
    N = 64
    g = torch.Generator().manual_seed(3)
    for step in (0, 5, 6, 30):
        logits = torch.randn(N, 3, generator=g)
        reg = torch.rand(N, 2, generator=g) * 1.1
        reg[0] = torch.tensor([1.0, 0.0]); reg[1] = torch.tensor([43.0 / 43.0, 0.5])
        unf = torch.rand(N, generator=g) > 0.2
        t, dl, unf2 = cape_ref.next_tokens(logits.argmax(-1), reg, unf.clone(), step, CFG)
        u = unf.to(torch.int32).to(DEV)
        tok = torch.empty(4, N, dtype=torch.int64, device=DEV)
        de = torch.empty(4, N, device=DEV)
        ops.decode_next_tokens(logits.to(DEV), reg.to(DEV), u, tok, de, torch.tensor([step], dtype=torch.int32, device=DEV),
                               N, 44, 6, CFG.eos, CFG.sep, CFG.pad)
        for i, k in enumerate(("11", "12", "21", "22")):
            assert torch.equal(tok[i].cpu(), t[k]), (step, k)
        for i in range(4):
            assert torch.equal(de[i].cpu(), dl[i]), (step, i)
        assert torch.equal(u.cpu().bool(), unf2)


@pytest.mark.parametrize("M,N,K,am,bm", [(300, 256, 256, 0, 0), (1000, 256, 1024, 0, 0), (640, 128, 256, 0, 1),
                                         (256, 384, 3000, 1, 1), (70, 260, 512, 0, 0)])
def test_gemm_bf16x3_split_accuracy(M, N, K, am, bm):
    """The bf16x3 mode (3 bf16 MFMAs per product, fp32 accumulate) must stay within ~2^-15 of an fp64 product
    relative to the magnitude sum(|a||b|); exact fp32 MFMA is ~1e-7 on the same measure."""
    a = rnd(M, K, seed=1) if am == 0 else rnd(K, M, seed=1)
    b = rnd(N, K, seed=2) if bm == 0 else rnd(K, N, seed=2)
    A64 = (a if am == 0 else a.t()).double()
    B64 = (b.t() if bm == 0 else b).double()
    ref = A64 @ B64
    mag = A64.abs() @ B64.abs()
    out = torch.zeros(M, N, device=DEV)
    errs = {}
    old = ops.get_gemm_precision()
    try:
        for name in ("f32", "bf16x3"):
            ops.set_gemm_precision(name)
            ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, a_mode=am, b_mode=bm)
            errs[name] = ((out.cpu().double() - ref).abs() / mag).max().item()
    finally:
        ops.set_gemm_precision(old)
    assert errs["f32"] < 2e-6, errs
    assert errs["bf16x3"] < 4e-5, errs


@pytest.mark.parametrize("Mo,Ni,Kr,sk", [(256, 256, 3000, 8), (70, 256, 300, 1), (1024, 256, 6400, 16), (260, 36, 100, 1), (384, 256, 5000, 3)])
@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_gemm_wgrad_fused_bias_sums(Mo, Ni, Kr, sk, prec):
    """wgrad GEMM with the fused bias gradient: colsum_out[m] += sum_k A[m][k] (A = dY stored [tokens][out])."""
    dy, x = rnd(Kr, Mo, seed=1), rnd(Kr, Ni, seed=2)
    old = ops.get_gemm_precision()
    try:
        ops.set_gemm_precision(prec)
        out = torch.zeros(Mo, Ni, device=DEV)
        cs = torch.full((Mo,), 2.0, device=DEV)
        ops.gemm(dy.to(DEV), x.to(DEV), out, Mo, Ni, Kr, a_mode=1, b_mode=1, accumulate=True, split_k=sk, colsum_out=cs)
    finally:
        ops.set_gemm_precision(old)
    close(out, dy.t() @ x, tol=2e-4, name="tn")
    close(cs, 2.0 + dy.sum(0), tol=2e-4, name="fused bias sums")


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("M", [6400, 130, 20])        # 20 rows: the skinny (M <= 64) kernel
def test_gemm_column_limited_residual_and_batched_bias(prec, M):
    """Round 3 epilogue forms of the decoder's self-attention node: (a) one product over stacked weight rows with a residual of
    its own leading dimension on the first `res_cols` columns only (q | k | v = tgt [Wq; Wk; Wv]^T, `+ query_pos` on q);
    (b) a batch of products that read column blocks of one matrix, each with its own weight block and bias block (in_proj)."""
    C = 256
    x, w, pos = rnd(M, C, seed=1), rnd(3 * C, C, seed=2, scale=C ** -0.5), rnd(M, C, seed=3)
    in_w, in_b = rnd(3 * C, C, seed=4, scale=C ** -0.5), rnd(3 * C, seed=5)
    old = ops.get_gemm_precision()
    try:
        ops.set_gemm_precision(prec)
        xd, wd, pd = x.to(DEV), torch.nn.Parameter(w.to(DEV)), pos.to(DEV)
        qkv1 = torch.empty(M, 3 * C, device=DEV)
        ops.gemm(xd, wd, qkv1, M, 3 * C, C, residual=pd, ldr=C, res_cols=C)                    # register-stationary kernel (packed weights)
        ref1 = x @ w.t()
        ref1[:, :C] += pos
        close(qkv1, ref1, tol=1e-4, name="stacked product, residual on the first block")
        qkv1b = torch.empty(M, 3 * C, device=DEV)
        ops.gemm(xd, wd.detach()[:, :252].contiguous(), qkv1b, M, 3 * C, 252, lda=C, residual=pd, ldr=C, res_cols=C)   # tiled kernel (K = 252)
        ref1b = x[:, :252] @ w[:, :252].t()
        ref1b[:, :C] += pos
        close(qkv1b, ref1b, tol=1e-4, name="tiled kernel, residual on the first block")
        qkv2 = torch.empty(M, 3 * C, device=DEV)
        ops.gemm(qkv1, in_w.to(DEV), qkv2, M, C, C, lda=3 * C, ldb=C, ldc=3 * C, bias=in_b.to(DEV),
                 batch=(3, 3, 0, C, 0, C * C, 0, C), bias_strides=(0, C))
        r1 = qkv1.cpu()
        ref2 = torch.cat([r1[:, i * C:(i + 1) * C] @ in_w[i * C:(i + 1) * C].t() + in_b[i * C:(i + 1) * C] for i in range(3)], 1)
        close(qkv2, ref2, tol=1e-4, name="batch-3 in_proj with per-block bias")
        d1 = torch.empty(M, 3 * C, device=DEV)
        ops.gemm(qkv2, in_w.to(DEV), d1, M, C, C, a_mode=0, b_mode=1, lda=3 * C, ldb=C, ldc=3 * C, batch=(3, 3, 0, C, 0, C * C, 0, C))
        r2 = qkv2.cpu()
        close(d1, torch.cat([r2[:, i * C:(i + 1) * C] @ in_w[i * C:(i + 1) * C] for i in range(3)], 1), tol=1e-4, name="batch-3 data gradient")
    finally:
        ops.set_gemm_precision(old)


def test_decoder_self_attention_node_matches_the_chain_of_nodes():
    """hip/functional.DecSelfAttnFn against the round-2 composition linear x3 -> mha (same kernels underneath, different
    grouping): outputs and every gradient, dropout off."""
    from cape_amd.hip import functional as HF
    torch.manual_seed(3)
    N, L, C = 3, 200, 256
    mk = lambda *s, sc=0.06: torch.nn.Parameter(torch.randn(*s, device=DEV) * sc)
    wqkv = mk(3 * C, C)
    wq, wk, wv = (torch.nn.Parameter(wqkv.detach()[i * C:(i + 1) * C]) for i in range(3))      # views of one slab: adjacent, as in the arena
    in_w, in_b, out_w, out_b = mk(3 * C, C), mk(3 * C, sc=0.1), mk(C, C), mk(C, sc=0.1)
    tgt = torch.randn(N, L, C, device=DEV, requires_grad=True)
    pos = torch.randn(N, L, C, device=DEV, requires_grad=True)
    go = torch.randn(N, L, C, device=DEV)
    params = [wq, wk, wv, in_w, in_b, out_w, out_b]

    def grads(fn):
        for t_ in params + [tgt, pos]:
            t_.grad = None
        fn().backward(go)
        HF.Runtime.join()
        return [t_.grad.clone() for t_ in [tgt, pos] + params]

    def chain():
        q = HF.linear(tgt, wq, residual=pos)
        return HF.mha(q, HF.linear(tgt, wk), HF.linear(tgt, wv), in_w, in_b, out_w, out_b, 8, mask_mode=1)

    node = lambda: HF.dec_self_attn(tgt, pos, wq, wk, wv, in_w, in_b, out_w, out_b, 8)
    with torch.no_grad():
        close(node(), chain(), tol=1e-4, name="self-attention node forward")
    for i, (a, b) in enumerate(zip(grads(node), grads(chain))):
        close(a, b, tol=3e-4, name=f"self-attention node grad {i}")


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_group_dense_wgrads(prec, tile):
    """cape_gemm_group_f32: several weight-gradient products (different shapes, k-splits, with / without fused bias sums, ragged K)
    in ONE launch == the same products one by one on the CPU.  Destinations start non-zero: the group accumulates."""
    from cape_amd.hip import lib
    cases = [(256, 256, 6400, 4, True), (768, 256, 544, 1, True), (384, 256, 3000, 8, False), (68, 36, 100, 1, True),
             (1024, 256, 2050, 2, True), (256, 1024, 640, 3, False), (132, 260, 40, 1, False)]
    old = ops.get_gemm_precision()
    try:
        ops.set_gemm_precision(prec)
        descs, shapes, keep, refs = [], [], [], []
        sink = lambda d, k, sh: (descs.append(d), keep.append(k), shapes.append(sh))
        for i, (Mo, Ni, Kr, sk, cs) in enumerate(cases):
            dy, x = rnd(Kr, Mo, seed=10 + i), rnd(Kr, Ni, seed=30 + i)
            out = torch.full((Mo, Ni), 0.5, device=DEV)
            csum = torch.full((Mo,), 2.0, device=DEV) if cs else None
            ops._wgrad_sink[0] = sink
            try:
                ops.gemm(dy.to(DEV), x.to(DEV), out, Mo, Ni, Kr, a_mode=1, b_mode=1, accumulate=True, split_k=sk, colsum_out=csum)
            finally:
                ops._wgrad_sink[0] = None
            refs.append((out, 0.5 + dy.t() @ x, csum, 2.0 + dy.sum(0)))
        assert len(descs) == len(cases)                      # every product was queued, none launched
        ops.gemm_group(descs, shapes, tile)
    finally:
        ops.set_gemm_precision(old)
    for i, (out, ref, csum, cref) in enumerate(refs):
        close(out, ref, tol=2e-4, name=f"group item {i}")
        if csum is not None:
            close(csum, cref, tol=2e-4, name=f"group item {i} bias sums")
    # the planner's k-splits keep >= 8 k-tiles per block and never exceed the depth
    sp = ops.plan_group_splits([(256, 256, 6400)] * 16 + [(384, 256, 43520)], 64)
    assert all(1 <= s <= 64 for s in sp) and sp[-1] >= sp[0]
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm_group(descs * 5, shapes * 5, tile)          # 35 items > CAPE_GEMM_GROUP_MAX


def test_gemm_group_conv_wgrads():
    """Grouped im2col weight gradients (b_mode 3) of different geometries against F.conv2d's autograd."""
    geoms = [(2, 16, 16, 64, 64, 3, 1, 1), (2, 17, 15, 32, 48, 3, 2, 1), (2, 32, 32, 4, 64, 7, 2, 3), (1, 8, 8, 256, 256, 3, 2, 1)]
    descs, shapes, keep, refs = [], [], [], []
    sink = lambda d, k, sh: (descs.append(d), keep.append(k), shapes.append(sh))
    for i, (N, H, W, C, O, k, stride, pad) in enumerate(geoms):
        x, w, geom, OH, OW = _conv_case(N, H, W, C, O, k, stride, pad, seed=50 + i)
        w.requires_grad_(True)
        y = F.conv2d(x, w, stride=stride, padding=pad)
        g = rnd(*y.shape, seed=70 + i)
        y.backward(g)
        xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        gn = g.permute(0, 2, 3, 1).contiguous().to(DEV)
        dw = torch.zeros(O, k * k * C, device=DEV)
        ops._wgrad_sink[0] = sink
        try:
            ops.gemm(gn, xn, dw, O, k * k * C, N * OH * OW, a_mode=1, b_mode=3, lda=O, conv=geom, accumulate=True, split_k=1)
        finally:
            ops._wgrad_sink[0] = None
        refs.append((dw.view(O, k, k, C), w.grad.permute(0, 2, 3, 1)))
    for d, sk in zip(descs, ops.plan_group_splits([sh[:3] for sh in shapes], 64)):
        d.split_k = sk
    ops.gemm_group(descs, shapes, 64)
    for i, (dw, ref) in enumerate(refs):
        close(dw, ref, tol=2e-4, name=f"conv wgrad group item {i}")


def test_add_n_rows_strided_sources():
    """cape_add_n_rows_f32: a summand may be a column block of a wider buffer (the q part of the decoder's [dq | dk | dv] gradient)."""
    wide = rnd(3, 50, 768, seed=1).to(DEV)
    a, b, c = wide[..., :256], rnd(3, 50, 256, seed=2).to(DEV), wide[..., 512:]
    close(ops.add_n_rows([a, b, c]), (a + b + c).cpu(), tol=1e-6, name="add_n_rows")
    from cape_amd.hip import functional as HF
    x = torch.randn(3, 50, 256, device=DEV, requires_grad=True)
    u, v = HF.fanout(x, 2)
    g = torch.randn(3, 50, 768, device=DEV)
    torch.autograd.backward([u, v], [g[..., 256:512], g[..., :256].contiguous()])
    close(x.grad, (g[..., 256:512] + g[..., :256]).cpu(), tol=1e-6, name="fan-in of a strided gradient")


def test_gradient_slots_match_summation_pass(monkeypatch):
    """hip/functional._Slot: consumers of a multiply-used tensor add their data gradients into ONE buffer (GEMM accumulate
    epilogue) instead of a summation pass over separate buffers.  A post-norm block with dropout (the LayerNorm backward offers
    the first buffer, the FFN / Linear data gradients accumulate into it) and a 3-consumer fan-out, slots on == slots off."""
    from cape_amd.hip import functional as HF
    torch.manual_seed(1)
    C = 256
    mk = lambda *s, sc=0.06: torch.nn.Parameter(torch.randn(*s, device=DEV) * sc)
    w1, b1, w2, b2, wv, bv = mk(1024, C), mk(1024, sc=0.1), mk(C, 1024, sc=0.03), mk(C, sc=0.1), mk(C, C), mk(C, sc=0.1)
    g1, be1 = torch.nn.Parameter(torch.ones(C, device=DEV)), torch.nn.Parameter(torch.zeros(C, device=DEV))
    x0 = torch.randn(3, 200, C, device=DEV)
    go = torch.randn(3, 200, C, device=DEV)
    params = [w1, b1, w2, b2, wv, bv, g1, be1]
    HF.Runtime.seed(77, torch.device(DEV))

    def run(slots):
        monkeypatch.setattr(HF, "_SLOTS", slots)
        x = x0.clone().requires_grad_(True)
        for p_ in params:
            p_.grad = None
        a, b, c = HF.fanout(x, 3)                                            # value-projection-like consumer + FFN + residual path
        v = HF.linear(a, wv, bv)
        h = HF.ffn(b, w1, b1, w2, b2, dropout_p=0.1, rng_stream=5)
        y = HF.add_layernorm(c, h, g1, be1, dropout_p=0.1, rng_stream=6)     # backward runs first: offers its dx
        (y + v).backward(go)
        HF.Runtime.join()
        return [x.grad.clone()] + [p_.grad.clone() for p_ in params]

    ref = run(False)
    got = run(True)
    for i, (a_, b_) in enumerate(zip(got, ref)):
        close(a_, b_, tol=2e-4, name=f"slot grad {i}")
        assert float(b_.abs().max()) > 0


def test_deferred_weight_gradients_match_immediate_launches():
    """hip/functional.Runtime.defer_wgrad: a backward pass whose weight gradients are queued and launched in groups produces the
    gradients of the same pass with one launch per product (a stack of Linear / FFN nodes writing straight into arena-style
    .grad tensors)."""
    from cape_amd.hip import functional as HF
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(256, 256, device=DEV) * 0.06) for _ in range(6)]
    bs = [torch.nn.Parameter(torch.randn(256, device=DEV) * 0.1) for _ in range(6)]
    w1, b1 = torch.nn.Parameter(torch.randn(1024, 256, device=DEV) * 0.06), torch.nn.Parameter(torch.zeros(1024, device=DEV))
    w2, b2 = torch.nn.Parameter(torch.randn(256, 1024, device=DEV) * 0.03), torch.nn.Parameter(torch.zeros(256, device=DEV))
    params = ws + bs + [w1, b1, w2, b2]
    x = torch.randn(4, 200, 256, device=DEV)

    def run(defer, group):
        for p in params:
            p.grad = torch.zeros_like(p)
        old = (HF.Runtime.direct_grad, HF.Runtime.defer_wgrad, HF.Runtime.wgrad_group)
        HF.Runtime.direct_grad, HF.Runtime.defer_wgrad, HF.Runtime.wgrad_group = True, defer, group
        try:
            h = x
            for w, b in zip(ws, bs):
                h = HF.linear(h, w, b, relu=True)
            h = HF.ffn(h, w1, b1, w2, b2)
            h.square().mean().backward()
            HF.Runtime.join()
            torch.cuda.synchronize()
            assert HF.Runtime.wq_total == 0 and not HF.Runtime.wq_notify
        finally:
            HF.Runtime.direct_grad, HF.Runtime.defer_wgrad, HF.Runtime.wgrad_group = old
        return [p.grad.clone() for p in params]

    ref = run(False, 24)
    for group in (3, 24):
        got = run(True, group)
        for i, (g, r) in enumerate(zip(got, ref)):
            close(g, r, tol=2e-4, name=f"deferred grad {i} (group {group})")
            assert float(r.abs().max()) > 0


# ------------------------------------------------------------------------------------------------
# fused decode step kernels (csrc/decode_step.hip, decode.hip: cape_decode_advance)
# ------------------------------------------------------------------------------------------------
def _ln(x, g, b):
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


@pytest.mark.parametrize("N,K,Nout", [(1, 256, 256), (2, 256, 768), (32, 256, 384), (64, 256, 1024), (32, 1024, 256), (5, 256, 256)])
def test_decode_linear_variants(N, K, Nout):
    """Column-split weight-streaming product with LayerNorm-on-load of the input and of the residual, `+ pos` after the
    norm, a second product on the first 256 columns, ReLU and three strided output segments (the q|k|v launch)."""
    x, w, b = rnd(N, K, seed=1), rnd(Nout, K, seed=2, scale=K ** -0.5), rnd(Nout, seed=3)
    g1, b1 = rnd(K, seed=4).abs() + 0.5, rnd(K, seed=5) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    # plain + relu
    out = torch.empty(N, Nout, device=DEV)
    ops.decode_linear(xd, wd, [out], bias=bd, relu=True)
    close(out, F.relu(x.double() @ w.double().t() + b).float(), tol=2e-6, name="plain relu")
    if K <= 256:
        add = rnd(N, K, seed=6)
        ops.decode_linear(xd, wd, [out], bias=bd, in_ln=(g1.to(DEV), b1.to(DEV)), in_add=add.to(DEV))
        want = (_ln(x.double(), g1.double(), b1.double()) + add.double()) @ w.double().t() + b
        close(out, want.float(), tol=5e-6, name="ln-on-load + add")
        # broadcast rows (stride 0): the layer-0 query position embedding is one row for all images
        row = rnd(1, K, seed=7)
        ops.decode_linear(xd, wd, [out], in_add=row.to(DEV).expand(N, K))
        close(out, ((x + row).double() @ w.double().t()).float(), tol=2e-6, name="broadcast add")
    # residual with its own LayerNorm (rows of the model width)
    r = rnd(N, Nout, seed=8)
    if Nout == 256:
        g2, b2 = rnd(Nout, seed=9).abs() + 0.5, rnd(Nout, seed=10) * 0.1
        ops.decode_linear(xd, wd, [out], bias=bd, res=r.to(DEV), res_ln=(g2.to(DEV), b2.to(DEV)))
        close(out, (x.double() @ w.double().t() + b + _ln(r.double(), g2.double(), b2.double())).float(), tol=5e-6, name="ln residual")
    ops.decode_linear(xd, wd, [out], res=r.to(DEV))
    close(out, (x.double() @ w.double().t() + r).float(), tol=2e-6, name="raw residual")
    if Nout == 768:
        # q | k | v: second product on the first 256 columns, k / v into row 3 of a (N, 7, 256) cache
        x2, w2 = rnd(N, 256, seed=11), rnd(256, 256, seed=12, scale=1 / 16)
        q = torch.empty(N, 256, device=DEV)
        kc, vc = torch.full((N, 7, 256), 9.0, device=DEV), torch.full((N, 7, 256), 9.0, device=DEV)
        ops.decode_linear(xd, wd, [q, kc[:, 3], vc[:, 3]], bias=bd, X2=x2.to(DEV), W2=w2.to(DEV))
        full = x.double() @ w.double().t() + b
        full[:, :256] += x2.double() @ w2.double().t()
        close(q, full[:, :256].float(), tol=2e-6, name="q segment")
        close(kc[:, 3], full[:, 256:512].float(), tol=2e-6, name="k segment")
        close(vc[:, 3], full[:, 512:].float(), tol=2e-6, name="v segment")
        assert float((kc[:, 2] - 9).abs().sum()) == 0 and float((vc[:, 4] - 9).abs().sum()) == 0


@pytest.mark.parametrize("N,last", [(1, False), (32, False), (32, True), (3, True)])
def test_decode_tail(N, last):
    """LN3 -> coords MLP -> refinement -> (class head) -> next layer's query position embedding and level-scaled points."""
    L = 4
    p4 = rnd(N, 256, seed=1)
    g3, b3 = rnd(256, seed=2).abs() + 0.5, rnd(256, seed=3) * 0.1
    W1, B1, W2, B2 = rnd(256, 256, seed=4, scale=1 / 16), rnd(256, seed=5) * 0.1, rnd(256, 256, seed=6, scale=1 / 16), rnd(256, seed=7) * 0.1
    W3, B3 = rnd(2, 256, seed=8, scale=1 / 16), rnd(2, seed=9) * 0.1
    ref = torch.rand(N, 2, generator=torch.Generator().manual_seed(10))
    ref[0] = torch.tensor([0.0, 1.0])                                   # the eps clamps of inverse_sigmoid
    Wc, Bc = rnd(3, 256, seed=11, scale=1 / 16), rnd(3, seed=12)
    Wp, Bp = rnd(256, 256, seed=13, scale=1 / 16), rnd(256, seed=14) * 0.1
    gp, bp = rnd(256, seed=15).abs() + 0.5, rnd(256, seed=16) * 0.1
    vr = torch.rand(N, L, 2, generator=torch.Generator().manual_seed(17)) * 0.5 + 0.5
    d = lambda t_: t_.to(DEV)
    ref_out = torch.full((N, 5, 2), 7.0, device=DEV)                    # a slot of an (N, T, 2) buffer
    cls_out = torch.full((N, 5, 3), 7.0, device=DEV)
    hs_out = torch.full((N, 5, 256), 7.0, device=DEV)
    qpos, refin = torch.empty(N, 256, device=DEV), torch.empty(N, L, 2, device=DEV)
    ops.decode_tail(d(p4), (d(g3), d(b3)), ((d(W1), d(B1)), (d(W2), d(B2)), (d(W3), d(B3))), d(ref), ref_out[:, 2], ops.dim_t(DEV),
                    vr=d(vr), cls_head=(d(Wc), d(Bc)) if last else None, cls_out=cls_out[:, 2] if last else None,
                    pos_trans=None if last else (d(Wp), d(Bp), d(gp), d(bp)), qpos_out=None if last else qpos,
                    refin_out=None if last else refin, hs_out=hs_out[:, 2] if last else None)
    t4 = _ln(p4, g3, b3)
    delta = F.linear(F.relu(F.linear(F.relu(F.linear(t4, W1, B1)), W2, B2)), W3, B3)
    new_ref = torch.sigmoid(delta + cape_ref.inverse_sigmoid(ref))
    close(ref_out[:, 2], new_ref, tol=2e-6, name="refined points")
    assert float((ref_out[:, 1] - 7).abs().sum()) == 0
    if last:
        close(cls_out[:, 2], F.linear(t4, Wc, Bc), tol=2e-6, name="class logits")
        close(hs_out[:, 2], t4, tol=2e-6, name="hs")
    else:
        want_q = _ln(F.linear(cape_ref.query_pos_sine(new_ref[:, None])[:, 0], Wp, Bp), gp, bp)
        close(qpos, want_q, tol=2e-5, name="next query pos")       # sin/cos of arguments up to 2 pi on the device
        close(refin, new_ref[:, None, :] * vr, tol=2e-6, name="level-scaled points")


def test_decode_advance_tokens_alive_and_embedding(proc_sd):
    N = 37
    g = torch.Generator().manual_seed(5)
    table = proc_sd["base_model.transformer.decoder.token_embed.weight"]
    for step in (0, 5, 6, 30):
        logits = torch.randn(N, 4, 3, generator=g)                      # slot 1 of an (N, 4, 3) buffer
        reg = torch.rand(N, 4, 2, generator=g) * 1.1
        unf = torch.rand(N, generator=g) > 0.2
        t, dl, unf2 = cape_ref.next_tokens(logits[:, 1].argmax(-1), reg[:, 1], unf.clone(), step, CFG)
        u = unf.to(torch.int32).to(DEV)
        tok = torch.empty(4, N, dtype=torch.int64, device=DEV)
        de = torch.empty(4, N, device=DEV)
        emb = torch.empty(N, 256, device=DEV)
        alive = torch.full((3,), -1, dtype=torch.int32, device=DEV)
        ld, rd = logits.to(DEV), reg.to(DEV)
        ops.decode_advance(ld[:, 1], rd[:, 1], u, tok, de, step, N, 44, 6, CFG.eos, CFG.sep, CFG.pad, table=table.to(DEV),
                           embed_out=emb, alive_out=alive[1:2])
        for i, k in enumerate(("11", "12", "21", "22")):
            assert torch.equal(tok[i].cpu(), t[k]), (step, k)
        for i in range(4):
            assert torch.equal(de[i].cpu(), dl[i]), (step, i)
        assert torch.equal(u.cpu().bool(), unf2)
        assert alive.cpu().tolist() == [-1, int(unf2.sum()), -1]
        want = cape_ref.seq_embed(proc_sd, {"seq11": t["11"], "seq12": t["12"], "seq21": t["21"], "seq22": t["22"],
                                            "delta_x1": dl[0], "delta_x2": dl[1], "delta_y1": dl[2], "delta_y2": dl[3]})
        close(emb, want, tol=1e-6, name="next-step embedding")
