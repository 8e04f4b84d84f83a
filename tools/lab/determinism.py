"""Run-to-run reproducibility of the training gradient on one GPU, and how much the model itself amplifies a last-bit
perturbation: (a) the same batch twice with the atomic k-splits on; (b) the same twice with them off; (c) k-splits off and the
input image perturbed by 1 ulp-scale noise.  Tells a race (a >> c) from arrival-order rounding (a ~ c).
    python tools/lab/determinism.py [--image_size 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cape_amd  # noqa: E402,F401
from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn  # noqa: E402
from cape_amd.datasets.synthetic import SyntheticEpisodes  # noqa: E402
from cape_amd.hip import functional as HF, ops  # noqa: E402
from cape_amd.models import build_model  # noqa: E402
from cape_amd.models.cape_model import build_cape_model  # noqa: E402
from cape_amd.models.train_cape_episodic import get_args_parser  # noqa: E402
from cape_amd.runtime.optimizer import ArenaAdamW  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_size", type=int, default=64)
    a = ap.parse_args()
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", str(a.image_size), "--dropout", "0.0"])
    torch.manual_seed(1)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).cuda().train()
    crit = crit.cuda()
    HF.Runtime.seed(5, torch.device("cuda"))
    opt = ArenaAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, max_norm=0.1)
    ds = SyntheticEpisodes(tok, 4, a.image_size, 9, 2, seed=3)
    b = episodic_collate_fn([ds[j] for j in range(4)])
    im, sc, sm = b["query_images"].cuda(), b["support_coords"].cuda(), b["support_masks"].cuda()
    tg, sk = {k: v.cuda() for k, v in b["query_targets"].items()}, b["support_skeletons"]

    def grads(images):
        opt.zero_grad()
        out = model(samples=images, support_coords=sc, support_mask=sm, targets=tg, skeleton_edges=sk)
        (crit(out, tg)["_total"]).backward()
        HF.Runtime.join()
        torch.cuda.synchronize()
        return [x.grad.clone() for x in opt.arenas]

    def rel(g0, g1):
        return max(((x - y).abs().max() / y.abs().max()).item() for x, y in zip(g0, g1))

    grads(im)
    for name, off in (("k-splits on ", False), ("k-splits off", True)):
        ops._NO_SPLIT_K = off
        g0 = grads(im)
        print(f"{name}: same batch twice: max |dg| / max |g| per arena = {rel(grads(im), g0):.3e}, thrice {rel(grads(im), g0):.3e}", flush=True)
    noise = im * (1.0 + 1.2e-7 * torch.randn_like(im))
    print(f"k-splits off, image perturbed by 1.2e-7 relative: {rel(grads(noise), g0):.3e}", flush=True)
    noise = im * (1.0 + 1e-5 * torch.randn_like(im))
    print(f"k-splits off, image perturbed by 1e-5 relative:   {rel(grads(noise), g0):.3e}", flush=True)


if __name__ == "__main__":
    main()
