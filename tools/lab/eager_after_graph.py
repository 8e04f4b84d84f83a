"""Does the eager training step get slower (host side) once a hipGraph of the step has been captured in the process?
Times eager steps before / after the capture and prints the caching allocator's device-allocation counters."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cape_amd  # noqa: E402,F401
from bench import make_batches  # noqa: E402


def main():
    import argparse
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.hip import functional as HF
    from cape_amd.hip import ops
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.runtime.graph_step import GraphedTrainStep
    from cape_amd.runtime.optimizer import ArenaAdamW
    device = torch.device("cuda")
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "256"])
    torch.autograd.set_multithreading_enabled(False)
    torch.manual_seed(1234)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).to(device).train()
    crit = crit.to(device)
    HF.Runtime.seed(1000, device)
    opt = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay, max_norm=args.clip_max_norm)
    batches = make_batches(tok, 16, 2, 256, 17, 4, seed=100, device=device)
    rng = HF.Runtime.get_rng(device)

    def step(i):
        b = batches[i % 4]
        rng.advance()
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"], targets=b["targets"], skeleton_edges=b["skeleton"])
        crit(out, b["targets"])["_total"].backward()
        opt.step()
        opt.zero_grad()

    def timed(label, n=6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            step(i)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print(f"{label}: enqueue {(t1 - t0) / n * 1e3:.2f} ms/step, wall {(t2 - t0) / n * 1e3:.2f} ms/step; device allocs {st['num_device_alloc']} "
              f"frees {st['num_device_free']} retries {st['num_alloc_retries']} reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB", flush=True)

    for i in range(3):
        step(i)
    timed("eager, fresh process")
    timed("eager, again")
    g = GraphedTrainStep(model, crit, opt, edge_capacity=2048, eager_steps=1)
    for i in range(4):
        b = batches[i % 4]
        g(b["images"], b["support_coords"], b["support_mask"], b["targets"], b["skeleton"])
    torch.cuda.synchronize()
    timed("eager, after a capture")
    timed("eager, after a capture (2)")
    ops.set_gemm_precision("f32")
    timed("eager f32")
    timed("eager f32 (2)")
    ops.set_gemm_precision("bf16x3")
    del g
    torch.cuda.empty_cache()
    timed("eager, graph dropped + empty_cache")
    timed("eager, graph dropped (2)")


if __name__ == "__main__":
    main()
