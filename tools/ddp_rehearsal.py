"""Two data-parallel ranks of the real model on ONE GPU (gloo carrying the CUDA gradient buckets) -- a rehearsal of
runtime/data_parallel.py with everything the single-rank tests cannot reach: buckets launched from the direct-to-arena
weight-gradient notifications, the side stream, the communication stream, finish().  The all-reduced mean gradient must
equal the gradient of the two ranks' batches accumulated locally.  (RCCL itself needs one GPU per rank; the driver's
multi-GPU bench is the first place it runs.)     python tools/ddp_rehearsal.py"""
import argparse
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # two executions of the same mathematics are compared at 2e-4: keep the forward k-splits' arrival-order rounding out
    # (the model amplifies 1e-7 to 5e-3, tools/lab/determinism.py); backward k-splits and the side stream stay on
    os.environ.setdefault("CAPE_DETERMINISTIC", "1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import cape_amd  # noqa: F401
    from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn
    from cape_amd.datasets.synthetic import SyntheticEpisodes
    from cape_amd.hip import functional as HF
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.runtime.data_parallel import EpisodeDataParallel
    from cape_amd.runtime.optimizer import ArenaAdamW
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "64", "--dropout", "0.0"])
    torch.manual_seed(1 + rank)                     # different init per rank: the broadcast must fix it
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).cuda().train()
    crit = crit.cuda()
    HF.Runtime.seed(5, torch.device("cuda"))
    opt = ArenaAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, max_norm=0.1)
    ddp = EpisodeDataParallel(model, opt, bucket_mb=8.0)
    ds = SyntheticEpisodes(tok, 4 * world, 64, 9, 2, seed=3)
    batches = []
    for r in range(world):
        b = episodic_collate_fn([ds[r * 4 + j] for j in range(4)])
        batches.append((b["query_images"].cuda(), b["support_coords"].cuda(), b["support_masks"].cuda(),
                        {k: v.cuda() for k, v in b["query_targets"].items()}, b["support_skeletons"]))

    def backward(b, scale):
        im, sc, sm, tg, sk = b
        out = model(samples=im, support_coords=sc, support_mask=sm, targets=tg, skeleton_edges=sk)
        (crit(out, tg)["_total"] * scale).backward()

    # step 1 = calibration (use counts are learnt, every bucket goes out in finish()); step 2 = steady state: buckets are launched
    # from the notifications DURING the backward pass, on the communication stream, while later gradients are still being written
    opt.zero_grad()
    backward(batches[rank], ddp.loss_scale)
    ddp.finish()
    assert ddp.stats["launched_before_finish"] == 0
    opt.zero_grad()
    backward(batches[rank], ddp.loss_scale)
    ddp.finish()
    early = ddp.stats["launched_before_finish"]
    print(f"rank {rank}: steady-state step launched {early} of {len(ddp.buckets)} buckets before finish()", flush=True)
    assert early >= 1, "no bucket was launched from a notification during the backward pass"
    torch.cuda.synchronize()
    got = [a.grad.clone() for a in opt.arenas]
    # local reference: both ranks' batches accumulated without any exchange
    opt.zero_grad()
    with ddp.no_sync():
        for r in range(world):
            backward(batches[r], 1.0 / world)
    HF.Runtime.join()
    torch.cuda.synchronize()
    ok = True
    for g, a in zip(got, opt.arenas):
        want = a.grad
        err = (g - want).abs().max().item()
        scale = want.abs().max().item()
        print(f"rank {rank}: arena of {a.numel} floats: max |allreduced - accumulated| = {err:.3e} (max |grad| {scale:.3e})", flush=True)
        ok = ok and err <= 2e-4 * max(scale, 1e-6)
    if not ok or os.environ.get("CAPE_REHEARSAL_VERBOSE"):
        flat = {id(a): (g, a) for g, a in zip(got, opt.arenas)}
        rows = []
        for name, p_ in model.named_parameters():
            if p_.grad is None:
                continue
            for g, a in flat.values():
                off = (p_.grad.data_ptr() - a.grad.data_ptr()) // 4
                if 0 <= off < a.numel and p_.grad.data_ptr() >= a.grad.data_ptr():
                    d = (g[off:off + p_.numel()] - a.grad[off:off + p_.numel()]).abs().max().item()
                    rows.append((d, a.grad[off:off + p_.numel()].abs().max().item(), name))
        for d, m, name in sorted(rows, reverse=True)[:12]:
            print(f"rank {rank}:   {name}: err {d:.3e} (max |grad| {m:.3e})", flush=True)
    assert len(ddp.buckets) >= 4, len(ddp.buckets)
    assert ok, "gradient mismatch"
    # ---- the captured step under data parallelism (runtime/graph_step.py with ddp=): two graphs around allreduce_all().  Every
    # rank feeds ITS batch; after eager -> capture + replay -> replay the replicas must still hold identical parameters, and the
    # replayed step must have moved them like an eager data-parallel step from the same state does
    from cape_amd.runtime.graph_step import GraphedTrainStep
    opt.zero_grad()
    gstep = GraphedTrainStep(model, crit, opt, loss_scale=ddp.loss_scale, edge_capacity=256, eager_steps=1, ddp=ddp)
    im, sc, sm, tg, sk = batches[rank]
    snap = [a.data.clone() for a in opt.arenas]
    steps_before = ddp.stats["steps"]
    for i in range(3):
        losses = gstep(im, sc, sm, tg, sk)
        if os.environ.get("CAPE_REHEARSAL_STEP_SYNC"):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for ai, a in enumerate(opt.arenas):
        both = [torch.zeros_like(a.data) for _ in range(world)]
        dist.all_gather(both, a.data)
        d = (both[0] - both[1]).abs()
        print(f"rank {rank}: after 3 graphed steps: arena {ai} max |param(rank0) - param(rank1)| = {float(d.max()):.3e} ({int((d > 0).sum())} of {d.numel()} differ)", flush=True)
        assert float(d.max()) == 0.0, "replicas diverged under the graphed data-parallel step"
    assert len(gstep.cache) == 1 and next(iter(gstep.cache.values())).graph2 is not None
    assert ddp.stats["steps"] == steps_before + 3 and torch.isfinite(losses["_total"]).all()
    moved = sum(float((a.data - s0).abs().sum()) for a, s0 in zip(opt.arenas, snap))
    assert moved > 0
    print(f"rank {rank}: graphed data-parallel step ok (3 steps, replicas identical, exposed comm of the last step "
          f"{ddp.exposed_comm_ms():.3f} ms)", flush=True)
    dist.barrier()
    if rank == 0:
        print(f"ddp rehearsal ok: {len(ddp.buckets)} buckets", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29653), nprocs=2, join=True)
