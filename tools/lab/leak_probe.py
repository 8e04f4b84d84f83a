"""Which tensors stay alive from one eager training step to the next?  Prints memory_allocated per step and the shapes of the
CUDA tensors that appear between two steps and are still alive after the next one."""
import collections
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cape_amd  # noqa: E402,F401
from bench import make_batches  # noqa: E402


def live():
    out = {}
    for o in gc.get_objects():
        try:
            if isinstance(o, torch.Tensor) and o.is_cuda:
                out[id(o)] = (tuple(o.shape), o.untyped_storage().data_ptr(), o.untyped_storage().nbytes())
        except Exception:
            pass
    return out


def main():
    import argparse
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.hip import functional as HF
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
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
    nb = int(os.environ.get("NB", "4"))
    batches = make_batches(tok, 16, 2, 256, 17, nb, seed=100, device=device)
    rng = HF.Runtime.get_rng(device)

    def step(i):
        b = batches[i % nb]
        rng.advance()
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"], targets=b["targets"], skeleton_edges=b["skeleton"])
        crit(out, b["targets"])["_total"].backward()
        opt.step()
        opt.zero_grad()

    for i in range(4):
        step(i)
        torch.cuda.synchronize()
        print(f"step {i}: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, "
              f"pending {len(HF.Runtime.pending)} wq {HF.Runtime.wq_total}", flush=True)
    gc.collect()
    a = live()
    step(4); torch.cuda.synchronize(); gc.collect()
    b = live()
    step(5); torch.cuda.synchronize(); gc.collect()
    c = live()
    print(f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB; live cuda tensors {len(a)} -> {len(b)} -> {len(c)}")
    new = [v for k, v in b.items() if k not in a and k in c]
    cnt = collections.Counter((s, n) for s, _, n in new)
    for (shape, nbytes), k in sorted(cnt.items(), key=lambda kv: -kv[0][1] * kv[1])[:25]:
        print(f"  x{k}  shape {shape}  storage {nbytes / 2**20:.1f} MiB")
    # who holds one of them?
    import types
    victims = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and id(o) in b and id(o) not in a and id(o) in c
               and tuple(o.shape) == (32, 32, 32, 512)][:2]
    for v in victims:
        print("victim", tuple(v.shape), "refcount", sys.getrefcount(v))
        for r in gc.get_referrers(v):
            if r is victims or isinstance(r, types.FrameType):
                continue
            desc = type(r).__name__
            if isinstance(r, dict):
                desc += " keys=" + str(list(r.keys())[:8])
            elif isinstance(r, (list, tuple)):
                desc += f" len={len(r)} types={[type(x).__name__ for x in r[:6]]}"
            print("   referrer:", desc)
            for r2 in gc.get_referrers(r)[:6]:
                if isinstance(r2, types.FrameType) or r2 is victims:
                    continue
                d2 = type(r2).__name__
                if isinstance(r2, dict):
                    d2 += " keys=" + str(list(r2.keys())[:8])
                elif isinstance(r2, (list, tuple)):
                    d2 += f" len={len(r2)}"
                print("        <-", d2)
    del victims
    for i in range(6, 10):
        step(i)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"after 10 steps: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB "
          f"device allocs {st['num_device_alloc']}")


if __name__ == "__main__":
    main()
