"""hipGraph capture of the episodic training step.

One optimizer step of the CAPE path is ~1400 kernel launches (715 GEMMs among them); launched one by one through
autograd + ctypes they cost ~30 ms of host time, about what the GPU needs to execute them.  The step has static shapes
for a fixed episode batch geometry, keeps its dropout seed / step counter on the device and decides nothing on the host,
so the whole of it -- forward, criterion, backward with the side-stream weight gradients, clip + AdamW, zero_grad -- is
captured once per input-shape signature (after `eager_steps` ordinary calls) and replayed; a replay costs the host three small copies and one launch.

    step = GraphedTrainStep(model, criterion, optimizer)
    losses = step(images, support_coords, support_mask, targets, skeleton_edges)    # dict of device scalars

Data parallel (`ddp=` an EpisodeDataParallel): the step is captured as TWO graphs -- (1) forward, criterion, backward with every
weight gradient landed in the arenas, (2) clip + AdamW + zero_grad + re-pack -- and the bucket all-reduces are enqueued between the
two replays (`EpisodeDataParallel.allreduce_all`): collectives stay outside the captures, the host does two replays and one
collective call per bucket instead of ~900 launches.  The exchange is then not overlapped with the backward pass (the eager step
overlaps it, at ~19 ms of host work per step); bench.py times both at N > 1 and reports the faster."""
import torch

from ..hip import functional as HF
from ..hip import ops
from ..models.graph_utils import DeviceSkeleton


def _sig(t):
    return (tuple(t.shape), str(t.dtype)) if isinstance(t, torch.Tensor) else repr(t)


class _Captured:
    def __init__(self, graph, static_in, static_targets, static_skel, losses, keep, graph2=None):
        self.graph, self.static_in, self.static_targets, self.static_skel = graph, static_in, static_targets, static_skel
        self.losses, self.keep, self.graph2 = losses, keep, graph2
        # the captured re-pack launch reads the packed-weight item table of this moment: keep that tensor alive, and remember
        # which registry it describes -- a weight registered later is not in it, the graph is then dropped and re-captured
        self.pack_table = ops.PackedWeights._table
        self.pack_sig = ops.PackedWeights.signature()


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, loss_scale=1.0, edge_capacity=None, max_graphs=8, eager_steps=2, ddp=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.ddp = ddp if (ddp is not None and ddp.world > 1) else None
        self.loss_scale, self.edge_capacity, self.max_graphs = float(loss_scale), edge_capacity, max_graphs
        self.eager_steps = eager_steps          # calls per shape signature that run eagerly before the capture
        self.cache, self.seen = {}, {}

    # ------------------------------------------------------------------------------------------------
    def _key(self, images, support_coords, support_mask, targets, n_edges):
        cap = self._capacity(n_edges)
        return (_sig(images), _sig(support_coords), _sig(support_mask), tuple((k, _sig(v)) for k, v in sorted(targets.items())), cap)

    def _capacity(self, n_edges):
        if self.edge_capacity is not None:
            if n_edges > self.edge_capacity:
                raise ValueError(f"{n_edges} skeleton edges exceed edge_capacity={self.edge_capacity}")
            return self.edge_capacity
        return max(64, 1 << (max(n_edges, 1) - 1).bit_length())       # power-of-two buckets keep the graph count small

    def _fwd_bwd(self, images, support_coords, support_mask, targets, skeleton):
        HF.Runtime.get_rng(images.device).advance()
        out = self.model(samples=images, support_coords=support_coords, support_mask=support_mask, targets=targets,
                         skeleton_edges=skeleton)
        losses = self.criterion(out, targets)
        (losses["_total"] * self.loss_scale).backward()
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in losses.items()}

    def _eager(self, images, support_coords, support_mask, targets, skeleton):
        losses = self._fwd_bwd(images, support_coords, support_mask, targets, skeleton)
        if self.ddp is not None:
            self.ddp.finish()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return losses

    def _capture(self, images, support_coords, support_mask, targets, skel_lists, cap):
        dev = images.device
        s_in = [images.clone(), support_coords.clone(), support_mask.clone()]
        s_tg = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in targets.items()}
        s_sk = DeviceSkeleton.from_lists(skel_lists, dev, capacity=cap)
        torch.cuda.synchronize()
        g, g2 = torch.cuda.CUDAGraph(), None
        HF.Runtime.capture_keep = []
        try:
            if self.ddp is None:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):    # other threads (pin-memory workers) stay free to call HIP
                    losses = self._eager(s_in[0], s_in[1], s_in[2], s_tg, s_sk)
            else:
                # two captures around the exchange: no hook launches a collective inside a capture (no_sync), every weight
                # gradient of the pass has joined the capture stream when graph 1 ends
                with self.ddp.no_sync():
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        losses = self._fwd_bwd(s_in[0], s_in[1], s_in[2], s_tg, s_sk)
                        HF.Runtime.join()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode="thread_local"):
                    self.optimizer.step()
                    self.optimizer.zero_grad()
            keep = HF.Runtime.capture_keep
        finally:
            HF.Runtime.capture_keep = None
        return _Captured(g, s_in, s_tg, s_sk, losses, keep, g2)

    # ------------------------------------------------------------------------------------------------
    def __call__(self, images, support_coords, support_mask, targets, skeleton_edges):
        if skeleton_edges is None:
            skeleton_edges = [[] for _ in range(support_coords.shape[0])]
        flat, start = DeviceSkeleton.flatten(skeleton_edges)
        key = self._key(images, support_coords, support_mask, targets, len(flat))
        c = self.cache.get(key)
        if c is None:
            n = self.seen.get(key, 0)
            if n < self.eager_steps:            # first calls run eagerly (allocator pools, lazily set kernel attributes)
                self.seen[key] = n + 1
                return self._eager(images, support_coords, support_mask, targets, skeleton_edges)
            if len(self.cache) >= self.max_graphs:
                self.cache.pop(next(iter(self.cache)))
            # capture records the step without executing it; the replay below is this call's one optimizer step
            c = self.cache[key] = self._capture(images, support_coords, support_mask, targets, skeleton_edges, key[-1])
            self._replay(c)
            return c.losses
        if c.pack_sig != ops.PackedWeights.signature():
            # a weight was registered (or dropped) after the capture: the captured table no longer covers the registry
            del self.cache[key]
            self.seen[key] = self.eager_steps
            return self(images, support_coords, support_mask, targets, skeleton_edges)
        c.static_in[0].copy_(images, non_blocking=True)
        c.static_in[1].copy_(support_coords, non_blocking=True)
        c.static_in[2].copy_(support_mask, non_blocking=True)
        for k, v in targets.items():
            if isinstance(v, torch.Tensor):
                c.static_targets[k].copy_(v, non_blocking=True)
        if flat:
            c.static_skel.edges[:len(flat)].copy_(torch.tensor(flat, dtype=torch.int32), non_blocking=False)
        c.static_skel.start.copy_(torch.tensor(start, dtype=torch.int32), non_blocking=False)
        self._replay(c)
        return c.losses

    def _replay(self, c):
        self.optimizer.sync_lr()                # the schedule's learning rates live on the device: uploaded here if they moved
        c.graph.replay()
        if c.graph2 is not None:
            self.ddp.allreduce_all()            # between the two replays: the gradient exchange, stream-ordered behind graph 1
            c.graph2.replay()
        # the replayed optimizer kernel rewrote the arenas and the replayed pack launch refreshed every registered plane: advance
        # the epoch that everything derived from the weights keys on (folded decode projections, decode graphs) and mark the
        # registry current for it -- ArenaAdamW.step() does this in Python only once, at capture
        ops.PackedWeights.mark_repacked()
