"""Which framework (aten) operators still run inside one training step, and from which line of the package: everything that
is not a launch of libcape_hip.so shows up here (device kernels, copies, fills, allocations are listed by operator name).
    python tools/aten_audit.py [--episodes 16] [--image_size 256]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cape_amd  # noqa: E402,F401
from bench import make_batches  # noqa: E402

PKG = os.path.join(ROOT, "category-agnostic-pose-estimation_amd")
NO_KERNEL = ("aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::detach", "aten::t", "aten::transpose", "aten::permute",
             "aten::expand", "aten::slice", "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::alias", "aten::reshape",
             "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::unbind", "aten::split", "aten::_reshape_alias",
             "aten::is_same_size", "aten::sym_size", "aten::sym_stride", "aten::sym_numel", "aten::sym_storage_offset", "aten::lift_fresh",
             "aten::new_empty", "aten::chunk", "aten::narrow", "aten::flatten", "aten::unflatten", "aten::view_as", "aten::new_empty_strided")


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0] if hasattr(func, "name") else str(func)
        if not name.startswith(NO_KERNEL):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=24)):
                if fr.filename.startswith(PKG) or fr.filename.endswith("bench.py"):
                    where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            self.counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=16)
    ap.add_argument("--image_size", type=int, default=256)
    a = ap.parse_args()
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.hip import functional as HF
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.runtime.optimizer import ArenaAdamW
    device = torch.device("cuda")
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", str(a.image_size)])
    torch.manual_seed(1234)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).to(device).train()
    crit = crit.to(device)
    HF.Runtime.seed(1000, device)
    opt = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay, max_norm=args.clip_max_norm)
    batches = make_batches(tok, a.episodes, 2, a.image_size, 17, 2, seed=100, device=device)
    rng = HF.Runtime.get_rng(device)

    def step():
        b = batches[0]
        rng.advance()
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                    targets=b["targets"], skeleton_edges=b["skeleton"])
        crit(out, b["targets"])["_total"].backward()
        opt.step()
        opt.zero_grad()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with Audit() as au:
        step()
    torch.cuda.synchronize()
    tot = sum(au.counts.values())
    print(f"{tot} framework operator calls in one training step (views / empty allocations not counted)")
    for (name, where), n in au.counts.most_common():
        print(f"{n:5d}  {name:32s} {where}")


if __name__ == "__main__":
    main()
