"""TEST INFRASTRUCTURE ONLY -- golden vectors for the bidirectional attention blocks.

Runs the reference's own `models/bixattn.py` classes (imported through oracle/refshim.py: timm is not installed, its
`Mlp` / `DropPath` come from the shim's restatement, so the MLP arithmetic is "parity unpinned" at the timm boundary) in
eval mode with procedural weights `procweights.tensor_for("bixattn.<block>.<param>", shape)` and writes
tests/golden/bixattn.npz.  Usage (build container only): python oracle/make_golden_bixattn.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import procweights, refshim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def fill(mod, prefix):
    sd = {k: procweights.tensor_for(prefix + "." + k, tuple(v.shape)) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd, strict=True)
    return mod.eval()


def inputs():
    """Seeded inputs, re-created by the tests (not stored): latents (2, 24, 256), patches (2, 280, 256) -- more than 256
    keys so that the kernel's key tiling is exercised."""
    rng = np.random.Generator(np.random.PCG64(31))
    lat = torch.from_numpy(rng.standard_normal((2, 24, 256)).astype(np.float32))
    pat = torch.from_numpy(rng.standard_normal((2, 280, 256)).astype(np.float32))
    return lat, pat


def main():
    refshim.install()
    from models.bixattn import BiXAttnBlock, CAOneSidedBlock
    lat, pat = inputs()
    blk = fill(BiXAttnBlock(256, 256, 256, 8, init_values=0.1), "bixattn.bi")
    blk0 = fill(BiXAttnBlock(256, 256, 256, 8, rv_bias=True, init_values=None), "bixattn.bi0")
    one = fill(CAOneSidedBlock(256, 256, 256, 8, init_values=0.1), "bixattn.one")
    with torch.no_grad():
        ol, op = blk(lat, pat)
        ol0, op0 = blk0(lat, pat)
        oo, none = one(lat, pat)
    assert none is None
    # patch outputs: every 5th row is kept (fixture size)
    np.savez_compressed(os.path.join(OUT, "bixattn.npz"), bi_lat=ol.numpy(), bi_pat=op[:, ::5].numpy(), bi0_lat=ol0.numpy(),
                        bi0_pat=op0[:, ::5].numpy(), one_lat=oo.numpy())
    print("wrote bixattn.npz")
    grads(blk, blk0, one, lat, pat)


def grads(blk, blk0, one, lat, pat):
    """Round 3: backward of the same three blocks (train mode, every drop rate 0 -- the reference's dropouts are then the
    identity), loss = <out_lat, c_lat> + <out_pat, c_pat> with seeded cotangents (re-created by the tests): input gradients in
    full (patches: every 5th row) and, per parameter, the gradient's L2 norm plus its first 8 elements."""
    rng = np.random.Generator(np.random.PCG64(32))
    c_lat = torch.from_numpy(rng.standard_normal((2, 24, 256)).astype(np.float32))
    c_pat = torch.from_numpy(rng.standard_normal((2, 280, 256)).astype(np.float32))
    out = {}
    for name, m in (("bi", blk), ("bi0", blk0), ("one", one)):
        m.train()
        m.zero_grad()
        xl, xp = lat.clone().requires_grad_(True), pat.clone().requires_grad_(True)
        ol, op = m(xl, xp)
        loss = (ol * c_lat).sum() + ((op * c_pat).sum() if op is not None else 0.0)
        loss.backward()
        out[name + "_dlat"] = xl.grad.numpy()
        out[name + "_dpat"] = xp.grad[:, ::5].numpy()
        names, norms, heads = [], [], []
        for k, p_ in m.named_parameters():
            if p_.grad is None:
                continue
            names.append(k)
            norms.append(float(p_.grad.norm()))
            heads.append(p_.grad.reshape(-1)[:8].numpy().copy())
        out[name + "_pnames"] = np.array(names)
        out[name + "_pnorms"] = np.array(norms, dtype=np.float64)
        out[name + "_pheads"] = np.stack(heads)
        m.eval()
    np.savez_compressed(os.path.join(OUT, "bixattn_grads.npz"), **out)
    print("wrote bixattn_grads.npz")


if __name__ == "__main__":
    main()
