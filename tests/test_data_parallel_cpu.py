"""World-size-2 `gloo` tests (CPU) of the episode data-parallel path: flat gradient arenas, reverse-order
buckets, post-accumulate hooks, no_sync accumulation, loss scaling -> mean gradient over ranks.
The optimizer kernel itself needs a GPU; here only the exchange is exercised."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.backbone = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Linear(16, 4))
        self.unused = torch.nn.Linear(3, 3)            # never receives a gradient

    def forward(self, x):
        return self.backbone(torch.relu(self.a(x)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cape_amd  # noqa: F401
    from cape_amd.runtime.data_parallel import EpisodeDataParallel
    from cape_amd.runtime.optimizer import ArenaAdamW
    torch.manual_seed(rank)                              # different init per rank: broadcast must fix it
    model = Toy()
    opt = ArenaAdamW(model, lr=1e-3, lr_backbone=1e-4)
    ddp = EpisodeDataParallel(model, opt, bucket_mb=0.0005)      # tiny buckets -> several per arena
    assert len(ddp.buckets) >= 3
    w0 = [p.detach().clone() for p in model.parameters()]
    gathered = [torch.zeros_like(opt.arenas[0].data) for _ in range(world)]
    dist.all_gather(gathered, opt.arenas[0].data)
    assert torch.equal(gathered[0], gathered[1]), "parameters not identical after broadcast"
    # two micro-batches of accumulation, exchange only on the boundary one
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank * 10 + i)) for i in range(2)]
    opt.zero_grad()
    with ddp.no_sync():
        (model(xs[0]).pow(2).mean() * ddp.loss_scale / 2).backward()
    (model(xs[1]).pow(2).mean() * ddp.loss_scale / 2).backward()
    ddp.finish()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    # reference: every rank recomputes both ranks' losses on a plain copy
    ref_model = Toy()
    ref_model.load_state_dict(model.state_dict())
    tot = 0
    for r in range(world):
        for i in range(2):
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + r * 10 + i))
            tot = tot + ref_model(x).pow(2).mean() / (2 * world)
    tot.backward()
    want = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in ref_model.named_parameters()}
    ok = all(torch.allclose(got[n], want[n], atol=1e-6) for n in want)
    # a second step must work after the pending counters were reset
    opt.zero_grad()
    (model(xs[0]).sum() * ddp.loss_scale).backward()
    ddp.finish()
    g2 = [torch.zeros_like(opt.arenas[1].grad) for _ in range(world)]
    dist.all_gather(g2, opt.arenas[1].grad)
    ok = ok and torch.equal(g2[0], g2[1])
    # ---- a parameter whose weight gradient is accumulated straight into the arena k times per backward (the decoder's
    # shared pos_trans: 6 uses) may release its bucket only after the k-th notification, not the first
    uw, ub = model.unused.weight, model.unused.bias          # never touched by autograd: stand-ins for direct-grad tensors
    bi = ddp._bucket_of[id(uw)]

    def run_step(check):
        opt.zero_grad()
        (model(xs[0]).sum() * ddp.loss_scale).backward()      # hooks of the autograd-accumulated members fire here
        for k in range(3):
            if check:
                assert bi not in ddp._launched, f"bucket launched after {k} of 3 uses"
            uw.grad.add_(1.0 + rank)                           # what a wgrad kernel would do
            ddp._on_direct_grad(uw)
            if k == 0:
                ub.grad.add_(2.0)
                ddp._on_direct_grad(ub)
        launched_early = bi in ddp._launched
        ddp.finish()
        return launched_early

    run_step(False)                                          # calibration: use counts are learned, buckets go in finish()
    ok = ok and ddp._uses[id(uw)] == 3 and ddp._uses[id(ub)] == 1
    members = ddp.buckets[bi][3]
    arena = ddp.buckets[bi][0]
    all_counted = all(id(arena.params[i]) in ddp._uses for i in members)
    early = run_step(True)
    ok = ok and (early == all_counted)                       # goes from the hook path once every member was seen
    want_uw = 3 * (1.0 + 0) / 1 + 3 * (1.0 + 1)              # sum over ranks of 3 adds each
    ok = ok and torch.allclose(uw.grad, torch.full_like(uw.grad, want_uw)) and torch.allclose(ub.grad, torch.full_like(ub.grad, 4.0))
    ok = ok and ddp.stats["steps"] == 4 and ddp.stats["bytes_per_step"] == 4 * sum(a.numel for a in opt.arenas)
    # ---- a notification beyond the calibrated use count arrives AFTER the bucket was all-reduced: hard error, not a
    # silent recount (both ranks raise at the same notification, no collective is left half-entered)
    opt.zero_grad()
    (model(xs[0]).sum() * ddp.loss_scale).backward()
    raised = False
    try:
        for k in range(4):
            uw.grad.add_(1.0)
            ddp._on_direct_grad(uw)
    except RuntimeError as e:
        raised = "calibration step counted 3 uses" in str(e)
    ok = ok and raised
    ub.grad.add_(2.0); ddp._on_direct_grad(ub)
    ddp.finish()                                             # the step still completes on both ranks (same counts everywhere)
    q.put((rank, ok, [n for n, _ in opt.dead]))
    dist.destroy_process_group()


def test_gradient_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_arena_keeps_shapes_strides_and_state_dict():
    sys.path.insert(0, ROOT)
    import cape_amd  # noqa: F401
    from cape_amd.runtime.arena import ParamGroupArena
    w = torch.nn.Parameter(torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last))
    b = torch.nn.Parameter(torch.randn(7))
    w0, b0 = w.detach().clone(), b.detach().clone()
    a = ParamGroupArena([("w", w), ("b", b)], "cpu")
    assert w.shape == (6, 4, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous() and torch.equal(w, w0) and torch.equal(b, b0)
    assert w.grad.shape == w.shape and w.grad.stride() == w.stride()
    a.grad.fill_(1.0)
    assert float(w.grad.sum()) == w.numel() and float(b.grad.sum()) == 7
    a.zero_grad()
    assert float(w.grad.abs().sum()) == 0
    (w.sum() * 2 + b.sum() * 3).backward()                 # autograd accumulates in place into the arena
    assert float(a.grad.sum()) == 2 * w.numel() + 3 * 7
    assert a.offsets[1] % 64 == 0


def _worker_logging(rank, world, port, q):
    """reduce_dict / MetricLogger / PCKEvaluator counters across two ranks (the only collectives besides
    the gradient exchange)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import cape_amd  # noqa: F401
    from cape_amd.util import misc as utils
    from cape_amd.util.eval_utils import PCKEvaluator
    ok = utils.get_world_size() == 2 and utils.get_rank() == rank and utils.is_main_process() == (rank == 0)
    red = utils.reduce_dict({"loss_ce": torch.tensor(1.0 + rank), "loss_coords": torch.tensor(10.0 * (rank + 1))})
    ok = ok and abs(float(red["loss_ce"]) - 1.5) < 1e-6 and abs(float(red["loss_coords"]) - 15.0) < 1e-6
    red_sum = utils.reduce_dict({"n": torch.tensor(3.0 + rank)}, average=False)
    ok = ok and float(red_sum["n"]) == 7.0
    ml = utils.MetricLogger()
    for v in ([1.0, 2.0] if rank == 0 else [3.0, 4.0, 5.0]):
        ml.update(loss=v)
    ml.synchronize_between_processes()
    ok = ok and abs(ml.meters["loss"].global_avg - 3.0) < 1e-6 and ml.meters["loss"].count == 5
    ev = PCKEvaluator(threshold=0.2)
    # rank 0: category 1, 2 of 3 visible keypoints correct; rank 1: category 1 (1 of 1) and category 2 (0 of 2)
    if rank == 0:
        pred = [np.array([[0.0, 0.0], [10.0, 10.0], [500.0, 500.0]])]
        gt = [np.array([[0.0, 0.0], [12.0, 10.0], [0.0, 0.0]])]
        ev.add_batch(pred, gt, [100.0], [100.0], category_ids=[1], visibility=[[2, 2, 2]])
    else:
        ev.add_batch([np.array([[5.0, 5.0]])], [np.array([[5.0, 6.0]])], [100.0], [100.0], category_ids=[1], visibility=[[2]])
        ev.add_batch([np.array([[0.0, 0.0], [0.0, 0.0]])], [np.array([[90.0, 90.0], [80.0, 80.0]])], [100.0], [100.0],
                     category_ids=[2], visibility=[[2, 1]])
    ev.synchronize_between_processes()
    r = ev.get_results()
    ok = ok and r["total_correct"] == 3 and r["total_visible"] == 6 and abs(r["pck_overall"] - 0.5) < 1e-9
    ok = ok and abs(r["mean_pck_categories"] - (0.75 + 0.0) / 2) < 1e-9
    q.put((rank, bool(ok), r))
    dist.destroy_process_group()


def test_logging_collectives_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_worker_logging, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2], "ranks disagree on the synchronised PCK"
