"""Does training behave the same across the arithmetic variants?  300 optimizer steps over 8 fixed synthetic batches (4 episodes x 2
queries, 256 x 256, dropout on, AdamW + clip as the CLI sets them) from the same initial weights, once per variant; prints the
loss every 50 steps.  Variants: exact-fp32 GEMMs + fp64 MSDA slab | bf16x3 + fixed-point slab (the default) | bf16x3 + fp64 slab |
stride-2 data gradients as one gather launch."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cape_amd  # noqa: E402,F401
from bench import make_batches  # noqa: E402


def run(name, precision, msda_accum, s2_classes, steps, device):
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.hip import functional as HF
    from cape_amd.hip import ops
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.runtime.optimizer import ArenaAdamW
    ops.set_gemm_precision(precision)
    ops.MSDA_VALUE_ACCUM = msda_accum
    HF._DGRAD_S2 = s2_classes
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "256"])
    torch.manual_seed(1234)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).to(device).train()
    opt = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay, max_norm=args.clip_max_norm)
    HF.Runtime.seed(7, device)
    batches = make_batches(tok, 4, 2, 256, 17, 8, 11, device)
    out = []
    acc = 0.0
    for i in range(steps):
        b = batches[i % len(batches)]
        o = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"], targets=b["targets"],
                  skeleton_edges=b["skeleton"])
        loss = crit(o, b["targets"])["_total"]
        loss.backward()
        opt.step(); opt.zero_grad()
        acc += float(loss.detach())
        if (i + 1) % 50 == 0:
            out.append(acc / 50)
            acc = 0.0
    print(f"{name:44s} mean loss per 50 steps: " + " ".join(f"{v:8.4f}" for v in out), flush=True)


def main():
    dev = torch.device("cuda")
    steps = int(os.environ.get("STEPS", "300"))
    run("f32 GEMMs, fp64 MSDA slab", "f32", "f64", True, steps, dev)
    run("bf16x3, fixed-point slab, s2 classes (default)", "bf16x3", "fx", True, steps, dev)
    run("bf16x3, fp64 slab", "bf16x3", "f64", True, steps, dev)
    run("bf16x3, fixed-point slab, one-launch s2 dgrad", "bf16x3", "fx", False, steps, dev)


if __name__ == "__main__":
    main()
