"""Where the host time of one eager training step goes: cProfile over two steps with the autograd engine kept on the calling
thread (so that the Python of the backward pass is visible).     python tools/host_profile.py"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cape_amd  # noqa: E402,F401
from bench import make_batches  # noqa: E402


def main():
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.hip import functional as HF
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.runtime.optimizer import ArenaAdamW
    device = torch.device("cuda")
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "256"])
    torch.manual_seed(1234)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).to(device).train()
    crit = crit.to(device)
    HF.Runtime.seed(1000, device)
    opt = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay, max_norm=args.clip_max_norm)
    batches = make_batches(tok, 16, 2, 256, 17, 2, seed=100, device=device)
    rng = HF.Runtime.get_rng(device)

    def step():
        b = batches[0]
        rng.advance()
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"], targets=b["targets"], skeleton_edges=b["skeleton"])
        crit(out, b["targets"])["_total"].backward()
        opt.step()
        opt.zero_grad()

    def timed(n=6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

    for _ in range(3):
        step()
    print("engine on its own thread : enqueue %.2f ms/step, wall %.2f ms/step" % timed(), flush=True)
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(2):
        step()
    print("engine on the caller     : enqueue %.2f ms/step, wall %.2f ms/step" % timed(), flush=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2):
        step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
