"""Episode data parallelism: one process per GPU, model replicated, episodes sharded by rank, ONE
exchange per optimizer step -- the gradient all-reduce (sum of per-rank grads of loss/world_size) over
RCCL/xGMI on the flat gradient arenas.  The reference has no distributed training; this
is the MI355X-native design for it:

  * gradients already live in contiguous slabs (runtime/arena.py), so a bucket is just a [lo, hi) range --
    no flatten/unflatten copies;
  * buckets are cut in *reverse* parameter order (~ the order backward produces them) at ~25 MB: on the
    fully-connected xGMI node RCCL moves 1/8 of a bucket over each of the 7 links concurrently, and 25 MB
    keeps each launch in the bandwidth regime while leaving several buckets to overlap with backward;
  * a bucket is launched on a side stream as soon as the last of its parameters has accumulated its
    gradient (post-accumulate-grad hooks, or -- for weight gradients that kernels accumulate straight into the arena --
    one notification per *use*: a parameter applied several times in the forward, e.g. the decoder's shared `pos_trans`,
    is complete only after as many notifications as the calibration step counted); `finish()` joins the stream before
    the optimizer step;
  * every rank must launch the same bucket sequence (collectives match by order): the sequence of the first steps is
    compared across ranks (`check_order_steps`), and `stats` reports bytes reduced and the exposed communication time;
  * micro-batches of gradient accumulation skip the exchange (`no_sync`), the boundary micro-batch reduces
    the accumulated sum;
  * parameters that never receive gradients are outside the arenas, identically on all ranks.
"""
import contextlib

import torch
import torch.distributed as dist

from ..hip import functional as HF


class EpisodeDataParallel:
    def __init__(self, model, optimizer, bucket_mb=25.0, process_group=None, check_order_steps=2):
        self.model, self.opt, self.pg = model, optimizer, process_group
        self.check_order_steps = check_order_steps      # compare the bucket launch order across ranks on the first steps
        self.steps_done = 0
        self.stats = {"steps": 0, "bytes_per_step": 0, "exposed_ms": 0.0, "launch_order": []}
        self._uses = {}             # id(param) -> notifications per backward (calibrated on the first exchanged step)
        self._calibrated = False
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.sync_enabled = True
        self.buckets = []           # (arena, lo, hi, [param indices])
        self._pending = {}
        self._handles = []
        self.comm_stream = torch.cuda.Stream() if (torch.cuda.is_available() and self.world > 1) else None
        if self.world > 1:
            self._broadcast_parameters()
            self._build_buckets(int(bucket_mb * (1 << 20) / 4))
            self._install_hooks()

    def _broadcast_parameters(self):
        """Rank 0's model everywhere: the arenas (all trainable tensors) in one message each, then whatever lives outside
        them -- frozen tensors (stem / layer1 of the backbone), the never-trained tensors of SURVEY fact 5 -- and the buffers."""
        in_arena = set()
        for a in self.opt.arenas:
            if a.numel:
                dist.broadcast(a.data, src=0, group=self.pg)
            in_arena.update(id(p) for p in a.params)
        for p in self.model.parameters():
            if id(p) not in in_arena:
                dist.broadcast(p.data, src=0, group=self.pg)
        for b in self.model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, src=0, group=self.pg)
        # the broadcasts wrote parameter memory behind autograd's version counters: everything derived from the weights
        # (packed GEMM planes, folded decode projections) follows the optimizer's epoch counter
        from ..hip import ops
        if torch.cuda.is_available():
            ops.PackedWeights.invalidate_and_repack()

    def _build_buckets(self, bucket_elems):
        for a in self.opt.arenas:
            n = len(a.params)
            hi_i = n
            while hi_i > 0:
                lo_i = hi_i - 1
                hi = a.offsets[hi_i] if hi_i < n else a.numel
                while lo_i > 0 and hi - a.offsets[lo_i - 1] <= bucket_elems:
                    lo_i -= 1
                self.buckets.append((a, a.offsets[lo_i], hi, list(range(lo_i, hi_i))))
                hi_i = lo_i

    def _install_hooks(self):
        self._bucket_of = {}
        for bi, (a, lo, hi, idxs) in enumerate(self.buckets):
            for i in idxs:
                p = a.params[i]
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
        # parameters whose wgrad kernels write the arena directly (no autograd accumulation) report here
        HF.Runtime.on_param_grad.append(self._on_direct_grad)
        self._reset_pending()

    def _on_direct_grad(self, p):
        """One call per weight-gradient launch into the arena.  A parameter used k times in the forward gets k launches;
        its bucket may go only after the k-th (the others are still accumulating into the same slot on the side stream)."""
        bi = self._bucket_of.get(id(p))
        if bi is None or not self.sync_enabled:
            return
        self._tick(p, bi)

    def _tick(self, p, bi):
        """One more gradient contribution of `p` has been enqueued (a direct-to-arena launch, or autograd's accumulation: a
        parameter may receive both kinds, in either order -- every contribution counts once)."""
        self._count[id(p)] = self._count.get(id(p), 0) + 1
        if not self._calibrated:
            return                                  # calibration step: count uses, buckets go in finish()
        need = self._uses.get(id(p))
        if need is None:
            self._calibrated = False                # a parameter the calibration step never saw: fall back to finish() and recount
            return
        if self._count[id(p)] > need:
            # its bucket went out at notification `need`; this launch accumulates an un-reduced contribution into a slot that the
            # all-reduce may be rewriting right now: the ranks would silently diverge
            raise RuntimeError(f"data parallel: a parameter of bucket {bi} received weight-gradient launch #{self._count[id(p)]} "
                               f"but the calibration step counted {need} uses per backward (data-dependent control flow in the "
                               "model?); its bucket has already been all-reduced")
        if self._count[id(p)] == need:
            self._param_done(bi)

    def _param_done(self, bi):
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _reset_pending(self):
        self._pending = {bi: len(b[3]) for bi, b in enumerate(self.buckets)}
        self._launched = set()
        self._seen = set()
        self._count = {}
        self._order = []
        self._early = 0

    def _make_hook(self, bi):
        def hook(_p):
            if not self.sync_enabled:
                return
            key = (bi, id(_p))
            if key in self._seen:
                return                              # autograd accumulates a parameter's gradient once per backward
            self._seen.add(key)
            self._tick(_p, bi)
        return hook

    def _launch(self, bi):
        if bi in self._launched:
            return
        self._launched.add(bi)
        self._order.append(bi)
        a, lo, hi, _ = self.buckets[bi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if HF.Runtime.side is not None:
                self.comm_stream.wait_stream(HF.Runtime.side)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(a.grad[lo:hi], group=self.pg)
        else:
            self._handles.append(dist.all_reduce(a.grad[lo:hi], group=self.pg, async_op=True))

    @contextlib.contextmanager
    def no_sync(self):
        old, self.sync_enabled = self.sync_enabled, False
        try:
            yield
        finally:
            self.sync_enabled = old

    def finish(self):
        """Call after the boundary micro-batch's backward: launches what hooks did not (parameters whose
        gradient was not produced this step still hold their zeros) and joins the communication."""
        if self.world < 2:
            return
        HF.Runtime.flush_wgrads()                   # queued weight-gradient groups and the notifications waiting for them
        self._early = len(self._order)              # buckets that went out during the backward pass (steady state: most of them)
        ev0 = ev1 = None
        if self.comm_stream is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev0.record()          # backward fully enqueued on the main stream
        for bi in range(len(self.buckets)):
            self._launch(bi)
        if self.comm_stream is not None:
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record(self.comm_stream)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for h in self._handles:
            h.wait()
        self._handles = []
        # use counts of this step become the expectation of the next ones (identical on every rank as long as no rank
        # takes a data-dependent branch; a mismatch shows up in the order check below)
        if not self._calibrated:
            self._uses = dict(self._count)
            self._calibrated = True
        elif any(self._count.get(k, 0) != v for k, v in self._uses.items()):
            self._uses, self._calibrated = dict(self._count), True
        order = list(self._order)
        self.stats["launched_before_finish"] = self._early
        if self.steps_done < self.check_order_steps:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, order, group=self.pg)
            if any(g != gathered[0] for g in gathered):
                raise RuntimeError(f"data-parallel ranks launched their gradient buckets in different orders: {gathered}")
        # every step (one small all-reduce of two integers): the ranks agree on how many notifications they saw and on a checksum
        # of the bucket order -- a rank that took a different path through the model shows up at the step where it happens
        chk = torch.tensor([sum(self._count.values()), sum((i + 1) * (b + 1) for i, b in enumerate(order))], dtype=torch.int64)
        lo, hi = chk.clone(), chk.clone()
        if self.comm_stream is not None and dist.get_backend(self.pg) == "nccl":
            lo, hi = lo.cuda(), hi.cuda()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo.cpu(), hi.cpu()):
            raise RuntimeError(f"data-parallel ranks disagree on this step's gradient notifications / bucket order "
                               f"(min {lo.tolist()}, max {hi.tolist()}, this rank {chk.tolist()})")
        self.steps_done += 1
        st = self.stats
        st["steps"] += 1
        st["bytes_per_step"] = sum(4 * (hi - lo) for _, lo, hi, _ in self.buckets)
        st["launch_order"] = order
        if ev0 is not None:
            st["_events"] = (ev0, ev1)              # read lazily: elapsed_time needs both events complete
        self._reset_pending()

    def allreduce_all(self):
        """Every bucket, now: the exchange of a step whose backward was replayed from a hipGraph (runtime/graph_step.py) -- no hook
        ran, the gradients of the whole pass are in the arenas when the replay has been enqueued.  Not overlapped with the
        backward (one graph), but the host side of the step shrinks to two replays and `len(buckets)` collective calls."""
        if self.world < 2:
            return
        ev0 = ev1 = None
        if self.comm_stream is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
        self._launched, self._order = set(), []
        for bi in range(len(self.buckets)):
            self._launch(bi)
        if self.comm_stream is not None:
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record(self.comm_stream)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for h in self._handles:
            h.wait()
        self._handles = []
        self.steps_done += 1
        st = self.stats
        st["steps"] += 1
        st["bytes_per_step"] = sum(4 * (hi - lo) for _, lo, hi, _ in self.buckets)
        st["launch_order"] = list(self._order)
        st["launched_before_finish"] = 0
        if ev0 is not None:
            st["_events"] = (ev0, ev1)
        self._reset_pending()

    def exposed_comm_ms(self):
        """Time the last step's communication ran past the end of its backward (0 = fully overlapped).  Synchronises."""
        ev = self.stats.get("_events")
        if ev is None:
            return 0.0
        ev[1].synchronize()
        return max(0.0, ev[0].elapsed_time(ev[1]))

    @property
    def loss_scale(self):
        """Multiply the loss by this so that the SUM all-reduce yields the mean gradient."""
        return 1.0 / self.world
