"""Which Python lines launch the torch-native (non-libcape) kernels of a training step: torch.profiler with stacks."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn  # noqa: E402
from cape_amd.datasets.synthetic import SyntheticEpisodes  # noqa: E402
from cape_amd.hip import functional as HF  # noqa: E402
from cape_amd.models import build_model  # noqa: E402
from cape_amd.models.cape_model import build_cape_model  # noqa: E402
from cape_amd.models.train_cape_episodic import get_args_parser  # noqa: E402
from cape_amd.runtime.optimizer import ArenaAdamW  # noqa: E402


def main():
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "128"])
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).cuda().train()
    crit = crit.cuda()
    opt = ArenaAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, max_norm=0.1)
    ds = SyntheticEpisodes(tok, 4, 128, 17, 2, seed=1)
    b = episodic_collate_fn([ds[j] for j in range(4)])
    im, sc, sm = b["query_images"].cuda(), b["support_coords"].cuda(), b["support_masks"].cuda()
    tg = {k: v.cuda() for k, v in b["query_targets"].items()}

    def step():
        HF.Runtime.get_rng(im.device).advance()
        out = model(samples=im, support_coords=sc, support_mask=sm, targets=tg, skeleton_edges=b["support_skeletons"])
        crit(out, tg)["_total"].backward()
        opt.step(); opt.zero_grad()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        step()
    torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        if ev.name.startswith("aten::") and ev.name.split("::")[1] in ("add", "add_", "copy_", "contiguous", "clone", "cat", "mul", "zeros", "zero_", "fill_", "sum", "to", "_to_copy", "stack", "where", "masked_fill", "index", "expand", "sub", "div", "neg", "cumsum", "eq", "ne", "bitwise_not", "bitwise_and", "bitwise_or", "any", "all", "index_select", "repeat", "arange", "full", "ones", "empty_like"):
            st = [s for s in ev.stack if "cape" in s or "category-agnostic" in s]
            where = st[0] if st else ("<autograd engine>" if not ev.stack else ev.stack[0])
            agg[(ev.name, where.strip()[-110:], str(ev.input_shapes)[:60])] += 1
    for (name, where, shp), c in agg.most_common(45):
        print(f"{c:4d}  {name:18s} {shp:60s} {where}")


if __name__ == "__main__":
    main()
