"""Autograd plumbing around the HIP kernels.

Each `torch.autograd.Function` here is glue: forward and backward only *launch* kernels of
libcape_hip.so (through `ops`) and keep the tensors backward needs.  No arithmetic of the hot path is
done by torch ops in this file.
"""
import ctypes
import os

import torch

from . import lib, ops


class Runtime:
    """Per-process runtime state shared by the Functions: dropout RNG state on device, and the
    weight-gradient sink: when parameters own gradient-arena views (runtime/arena.py) the wgrad kernels
    accumulate straight into them (GEMM epilogue `C += ...` / atomics) instead of materialising a tensor that
    autograd would add afterwards, and they run on a side stream so that they fill the idle CUs of the
    data-gradient chain's kernel tails.  `on_param_grad` callbacks let the data-parallel layer launch a
    bucket's all-reduce as soon as its gradients have been enqueued."""
    rng = None
    direct_grad = False
    use_side_stream = os.environ.get("CAPE_SIDE_STREAM", "1") == "1"
    side = None
    on_param_grad = []
    capture_keep = None          # list while a hipGraph capture is in progress (runtime/graph_step.py)

    pending = []                 # tensors read by enqueued side-stream kernels: kept alive until join()
    side_dirty = False           # work has been forked onto the side stream since the last join

    # Weight gradients are not launched where the backward pass produces them: nothing waits for one until the optimizer
    # step, so their GEMM descriptors are queued (per tile class) and submitted `wgrad_group` at a time as ONE grouped launch
    # (csrc/gemm_group.hip) -- and whatever is left when the autograd engine finishes the pass.  Round 2 paid 199 launches,
    # 24-64 k-splits each to fill the chip alone, for what is now ~15 launches of 4-8 k-splits.
    defer_wgrad = os.environ.get("CAPE_DEFER_WGRAD", "1") == "1"
    wgrad_group = int(os.environ.get("CAPE_WGRAD_GROUP", "12"))
    wq = {}                      # (tile, b_mode, precision) -> [(desc, keep, shape)]
    wq_total = 0
    wq_notify = []               # parameters whose notification (data-parallel bucket bookkeeping) waits for the queued products
    _final_cb = False

    @classmethod
    def side_stream(cls):
        if cls.side is None:
            cls.side = torch.cuda.Stream()
            cls.side_raw = cls.side.cuda_stream
        return cls.side

    @classmethod
    def join(cls):
        """Make the current stream wait for all enqueued side-stream work (call before the optimizer step)."""
        cls.flush_wgrads()
        if cls.side is not None and cls.side_dirty:     # (nothing forked since the last join: no event to wait for -- a capture that
            cls.side_dirty = False                      #  only holds the optimizer step must not touch the un-captured side stream)
            lib.call("cape_stream_join", ctypes.c_void_p(ops.raw_current_stream()), ctypes.c_void_p(cls.side_raw))
            if cls.capture_keep is None:
                cls.pending.clear()

    @classmethod
    def notify(cls, *params):
        if cls.wq_total or cls.wq_notify:               # behind queued products: the callbacks run when those have been launched
            cls.wq_notify.extend(p for p in params if p is not None)
            return
        for cb in cls.on_param_grad:
            for p in params:
                if p is not None:
                    cb(p)

    @classmethod
    def enqueue_wgrad(cls, desc, keep, shape):
        M, N, K, _, b_mode = shape
        key = (ops.group_tile(M, N, K), b_mode, desc.precision)
        cls.wq.setdefault(key, []).append((desc, keep, shape))
        cls.wq_total += 1
        if not cls._final_cb:
            try:                                        # whatever is still queued when this backward pass ends goes then
                torch.autograd.Variable._execution_engine.queue_callback(cls._on_backward_end)
                cls._final_cb = True
            except RuntimeError:                        # not inside a backward pass (a Function driven by hand): no deferral
                cls.flush_wgrads()
                return
        if cls.wq_total >= cls.wgrad_group or len(cls.wq[key]) >= lib.GEMM_GROUP_MAX:
            cls.flush_wgrads()

    @classmethod
    def _on_backward_end(cls):
        cls._final_cb = False
        cls.flush_wgrads()

    @classmethod
    def _launch_class(cls, key):
        items = cls.wq.get(key)
        if not items:
            return
        cls.wq[key] = []
        cls.wq_total -= len(items)
        tile = key[0]
        shapes = [it[2] for it in items]
        for it, sk in zip(items, ops.plan_group_splits([sh[:3] for sh in shapes], tile)):
            it[0].split_k = sk
        keep = [t for it in items for t in it[1] if t is not None]
        side = cls.use_side_stream
        if side:                                        # ordered after everything the main stream has been given so far
            cls.side_stream()
            lib.call("cape_stream_fork", ctypes.c_void_p(ops.raw_current_stream()), ctypes.c_void_p(cls.side_raw))
            cls.side_dirty = True
            ops._stream_override[0] = cls.side_raw
        try:
            for i in range(0, len(items), lib.GEMM_GROUP_MAX):
                ops.gemm_group([it[0] for it in items[i:i + lib.GEMM_GROUP_MAX]], shapes[i:i + lib.GEMM_GROUP_MAX], tile)
        finally:
            if side:
                ops._stream_override[0] = None
        if side:
            (cls.capture_keep if cls.capture_keep is not None else cls.pending).extend(keep)

    @classmethod
    def flush_wgrads(cls):
        """Launch every queued weight-gradient product (one grouped launch per tile class) and run the notifications
        that waited for them."""
        for key in list(cls.wq):
            cls._launch_class(key)
        if cls.wq_notify:
            ps, cls.wq_notify = cls.wq_notify, []
            for cb in cls.on_param_grad:
                for p in ps:
                    cb(p)

    @classmethod
    def get_rng(cls, device):
        if cls.rng is None:
            cls.rng = ops.RngState(0x5EED, device)
        return cls.rng

    @classmethod
    def seed(cls, seed, device):
        cls.rng = ops.RngState(int(seed), device)


def _mark_side_dirty():
    Runtime.side_dirty = True


ops._on_fork[0] = _mark_side_dirty


def capturing():
    """True while the current stream is being captured into a hipGraph: host decisions that would need a
    device->host sync take their conservative branch, and cross-stream lifetimes are handled by keeping tensors alive."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _c(t):
    return t if (t is None or t.is_contiguous()) else t.contiguous()


def _sink(p):
    """Gradient-arena view of parameter `p` (or of the parameter `p` is a plain view of), else None."""
    if not Runtime.direct_grad or p is None or not p.requires_grad:
        return None
    if isinstance(p, torch.nn.Parameter):
        return p.grad
    base = p._base
    if isinstance(base, torch.nn.Parameter) and base.grad is not None and base.requires_grad:
        off = p.storage_offset() - base.storage_offset() + base.grad.storage_offset()
        return torch.as_strided(base.grad, p.shape, p.stride(), off)
    return None


def _param_of(p):
    if isinstance(p, torch.nn.Parameter):
        return p
    return p._base if (p is not None and isinstance(p._base, torch.nn.Parameter)) else None


class _Slot:
    """Gradient slot of a tensor with several consumers (see `fanout`): the first consumer whose backward produces an exclusively
    owned data gradient leaves its buffer here; later consumers ADD theirs into it (GEMM `C += ...` epilogue) instead of writing
    a buffer of their own that a summation pass then reads again."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


_SLOTS = os.environ.get("CAPE_GRAD_SLOTS", "1") == "1"


def _slot_of(x):
    return getattr(x, "_cape_slot", None) if _SLOTS else None


def _grad_target(slot, shape, device):
    """(buffer viewed as `shape`, accumulate?) for a data gradient of `shape` (contiguous)."""
    if slot is not None and slot.buf is not None:
        b = slot.buf
        if b.is_contiguous() and b.numel() == int(torch.Size(shape).numel()) and b.device == device:
            return b.view(shape), True
    buf = torch.empty(shape, dtype=torch.float32, device=device)
    if slot is not None and slot.buf is None:
        slot.buf = buf
    return buf, False


def _slot_offer(slot, buf):
    """A non-GEMM producer (LayerNorm backward) offers its freshly written, exclusively owned gradient buffer."""
    if slot is not None and slot.buf is None and buf is not None and buf.is_contiguous():
        slot.buf = buf


class _Side:
    """with _Side(tensors...): kernels launched inside go to the side stream, ordered after the current stream's work so far
    (one C call: event record + wait; no framework stream switch); the listed tensors are kept alive until the next join --
    the caching allocator may otherwise hand their blocks to a later main-stream kernel while the side kernel still reads them."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        self.on = Runtime.use_side_stream
        self.defer = Runtime.defer_wgrad
        if self.defer:
            ops._wgrad_sink[0] = self._sink
        if not self.on:
            return self
        Runtime.side_stream()
        ops._stream_override[0] = Runtime.side_raw
        if self.defer:
            ops._lazy_fork[0] = True                    # queued products need no fork here; any other launch orders the stream first
        else:
            lib.call("cape_stream_fork", ctypes.c_void_p(ops.raw_current_stream()), ctypes.c_void_p(Runtime.side_raw))
            Runtime.side_dirty = True
        return self

    def _sink(self, desc, keep, shape):
        # the queue keeps the product's operands alive until its launch; on a capture nothing is ever released
        Runtime.enqueue_wgrad(desc, keep, shape)

    def __exit__(self, *a):
        ops._wgrad_sink[0] = None
        if not self.on:
            return False
        ops._stream_override[0] = None
        ops._lazy_fork[0] = False
        (Runtime.capture_keep if Runtime.capture_keep is not None else Runtime.pending).extend(self.tensors)
        if len(Runtime.pending) > 8192:                 # a caller that never joins (backward without an optimizer step)
            Runtime.join()
        return False


# ------------------------------------------------------------------------------------------------
# Linear (+bias, +relu, +dropout, +residual) -- F.linear and its autograd
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu, dropout_p, rng_stream):
        K = x.shape[-1]
        N = weight.shape[0]
        assert weight.shape[1] == K and weight.stride(1) == 1
        x2 = _c(x).view(-1, K)
        M = x2.shape[0]
        res2 = _c(residual).view(-1, N) if residual is not None else None
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        rng = Runtime.get_rng(x.device) if dropout_p > 0 else None
        ops.gemm(x2, weight, y, M, N, K, ldb=weight.stride(0), bias=bias, residual=res2, relu=relu, dropout_p=dropout_p,
                 rng=rng, rng_stream=rng_stream)
        ctx.save_for_backward(x2, weight, y if (relu or dropout_p > 0) else None)
        ctx.w_ref, ctx.b_ref = weight, bias
        ctx.slot = _slot_of(x)
        ctx.meta = (relu, dropout_p, bias is not None, residual is not None, x.shape, M, N, K)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        relu, p, has_bias, has_res, xshape, M, N, K = ctx.meta
        dy2 = _c(dy).view(M, N)
        dpre = dy2
        if relu or p > 0:
            assert relu, "dropout epilogue is only used together with relu"
            dpre = ops.relu_drop_bwd(dy2, y, 1.0 / (1.0 - p) if p > 0 else 1.0)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx, acc = _grad_target(ctx.slot, (M, K), dy.device)
            ops.gemm(dpre, weight, dx, M, K, N, a_mode=0, b_mode=1, ldb=weight.stride(0), accumulate=acc)
            dx = dx.view(xshape)
        wsink, bsink = _sink(ctx.w_ref), _sink(ctx.b_ref)
        need_w, need_b = ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        if (need_w and wsink is not None) or (need_b and bsink is not None):
            with _Side(dpre, x2):
                fuse_b = need_b and bsink is not None and need_w and wsink is not None   # bias sums ride in the wgrad
                if need_w and wsink is not None:
                    ops.gemm(dpre, x2, wsink, N, K, M, a_mode=1, b_mode=1, lda=N, ldb=K, ldc=wsink.stride(0), accumulate=True,
                             split_k=ops.pick_split_k(N, K, M), colsum_out=bsink if fuse_b else None)
                if need_b and bsink is not None and not fuse_b:
                    ops.colsum(dpre, M, N, bsink)
            Runtime.notify(_param_of(ctx.w_ref) if (need_w and wsink is not None) else None,
                           _param_of(ctx.b_ref) if (need_b and bsink is not None) else None)
        if need_w and wsink is None:
            dw = torch.zeros(N, K, dtype=torch.float32, device=dy.device)
            ops.gemm(dpre, x2, dw, N, K, M, a_mode=1, b_mode=1, lda=N, ldb=K, accumulate=True,
                     split_k=ops.pick_split_k(N, K, M))
        if need_b and bsink is None:
            db = torch.zeros(N, dtype=torch.float32, device=dy.device)
            ops.colsum(dpre, M, N, db)
        if has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None


def linear(x, weight, bias=None, residual=None, relu=False, dropout_p=0.0, rng_stream=0):
    return LinearFn.apply(x, weight, bias, residual, relu, float(dropout_p), int(rng_stream))


class FFNFn(torch.autograd.Function):
    """y = (dropout(relu(x W1^T + b1))) W2^T + b2 -- the transformer feed-forward pair as one node, so that the backward
    can let the dgrad of the second linear emit the pre-activation gradient of the first directly (GEMM epilogue gated by
    the saved hidden activation) instead of running a separate relu/dropout-backward pass over the (rows x 1024) tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, dropout_p, rng_stream):
        K, Hd, N = x.shape[-1], w1.shape[0], w2.shape[0]
        x2 = _c(x).view(-1, K)
        M = x2.shape[0]
        rng = Runtime.get_rng(x.device) if dropout_p > 0 else None
        h = torch.empty(M, Hd, dtype=torch.float32, device=x.device)
        ops.gemm(x2, w1, h, M, Hd, K, ldb=w1.stride(0), bias=b1, relu=True, dropout_p=dropout_p, rng=rng, rng_stream=rng_stream)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ops.gemm(h, w2, y, M, N, Hd, ldb=w2.stride(0), bias=b2)
        ctx.save_for_backward(x2, w1, w2, h)
        ctx.refs = (w1, b1, w2, b2)
        ctx.slot = _slot_of(x)
        ctx.meta = (dropout_p, x.shape, M, K, Hd, N)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h = ctx.saved_tensors
        p, xshape, M, K, Hd, N = ctx.meta
        dy2 = _c(dy).view(M, N)
        # d(pre-activation of linear1) = (dy W2) gated by the saved hidden activation
        dpre = torch.empty(M, Hd, dtype=torch.float32, device=dy.device)
        ops.gemm(dy2, w2, dpre, M, Hd, N, a_mode=0, b_mode=1, ldb=w2.stride(0), mask_src=h, mask_scale=1.0 / (1.0 - p) if p > 0 else 1.0)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, acc = _grad_target(ctx.slot, (M, K), dy.device)
            ops.gemm(dpre, w1, dx, M, K, Hd, a_mode=0, b_mode=1, ldb=w1.stride(0), accumulate=acc)
            dx = dx.view(xshape)
        sinks = [_sink(t) for t in ctx.refs]
        if all(k is not None for k in sinks) and all(ctx.needs_input_grad[1:5]):
            with _Side(dy2, dpre, h, x2):
                ops.gemm(dy2, h, sinks[2], N, Hd, M, a_mode=1, b_mode=1, lda=N, ldb=Hd, ldc=sinks[2].stride(0), accumulate=True,
                         split_k=ops.pick_split_k(N, Hd, M), colsum_out=sinks[3])
                ops.gemm(dpre, x2, sinks[0], Hd, K, M, a_mode=1, b_mode=1, lda=Hd, ldb=K, ldc=sinks[0].stride(0), accumulate=True,
                         split_k=ops.pick_split_k(Hd, K, M), colsum_out=sinks[1])
            Runtime.notify(*[_param_of(t) for t in ctx.refs])
            return dx, None, None, None, None, None, None
        dev = dy.device
        dw2 = torch.zeros(N, Hd, dtype=torch.float32, device=dev); db2 = torch.zeros(N, dtype=torch.float32, device=dev)
        dw1 = torch.zeros(Hd, K, dtype=torch.float32, device=dev); db1 = torch.zeros(Hd, dtype=torch.float32, device=dev)
        ops.gemm(dy2, h, dw2, N, Hd, M, a_mode=1, b_mode=1, lda=N, ldb=Hd, accumulate=True, split_k=ops.pick_split_k(N, Hd, M), colsum_out=db2)
        ops.gemm(dpre, x2, dw1, Hd, K, M, a_mode=1, b_mode=1, lda=Hd, ldb=K, accumulate=True, split_k=ops.pick_split_k(Hd, K, M), colsum_out=db1)
        return dx, dw1, db1, dw2, db2, None, None


_NO_FFN_FUSE = os.environ.get("CAPE_NO_FFN_FUSE") is not None
# CAPE_DETERMINISTIC=1: no atomic k-split in the forward pass (convolution forward), so activations are bitwise reproducible
# run to run.  The backward k-splits stay: the backward pass is linear in dY for fixed activations, their arrival-order
# rounding (~1e-6 of the gradient) is not amplified.  Measured (tools/lab/determinism.py, profiles/
# r02_determinism.txt): this model turns a 1.2e-7 relative perturbation of the input image into a 3e-3..7e-3 relative change
# of the gradient (discontinuous pieces: bilinear-sampling cell boundaries, L1 signs, ReLU gates), and the forward
# k-splits' arrival order does the same -- comparable to the bf16x3 / fp32 difference, invisible to training, but too
# large for tests that compare two executions of the same mathematics at 2e-4.
_DETERMINISTIC = os.environ.get("CAPE_DETERMINISTIC", "0") == "1"


def ffn(x, w1, b1, w2, b2, dropout_p=0.0, rng_stream=0):
    """linear2(dropout(relu(linear1(x)))) with both biases (deformable_transformer.py:95,99-100; deformable_transformer_v2.py:
    314-318; nn.TransformerEncoderLayer's feed-forward in geometric_support_encoder.py)."""
    if _NO_FFN_FUSE:          # tuning switch: two separate linear nodes (relu/dropout backward as its own pass)
        return linear(linear(x, w1, b1, relu=True, dropout_p=dropout_p, rng_stream=rng_stream), w2, b2)
    return FFNFn.apply(x, w1, b1, w2, b2, float(dropout_p), int(rng_stream))


def _adjacent(a, b):
    """b starts where a ends, in the same storage (both dense, row-major)."""
    return (a.is_contiguous() and b.is_contiguous() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size())


class LinearCat2Fn(torch.autograd.Function):
    """[x W1^T + b1 | x W2^T + b2]  (sampling offsets | attention logits of MSDeformAttn)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        K = x.shape[-1]
        N1, N2 = w1.shape[0], w2.shape[0]
        x2 = _c(x).view(-1, K)
        M = x2.shape[0]
        y = torch.empty(M, N1 + N2, dtype=torch.float32, device=x.device)
        # in the flat arenas the two weights (and the two biases) sit back to back (runtime/arena._colocate): one launch
        adjacent = _adjacent(w1, w2) and _adjacent(b1, b2)
        if adjacent:
            ops.gemm(x2, torch.as_strided(w1, (N1 + N2, K), (K, 1)), y, M, N1 + N2, K, bias=torch.as_strided(b1, (N1 + N2,), (1,)))
        else:
            ops.gemm(x2, w1, y, M, N1, K, bias=b1, ldc=N1 + N2)
            ops.gemm(x2, w2, y[:, N1:], M, N2, K, bias=b2, ldc=N1 + N2)
        ctx.save_for_backward(x2, w1, w2)
        ctx.refs = (w1, b1, w2, b2)
        ctx.meta = (x.shape, M, N1, N2, K, adjacent)
        return y.view(*x.shape[:-1], N1 + N2)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2 = ctx.saved_tensors
        xshape, M, N1, N2, K, adjacent = ctx.meta
        NT = N1 + N2
        dy2 = _c(dy).view(M, NT)
        dev = dy.device
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            ops.gemm(dy2, w1, dx, M, K, N1, a_mode=0, b_mode=1, lda=NT)
            ops.gemm(dy2[:, N1:], w2, dx, M, K, N2, a_mode=0, b_mode=1, lda=NT, accumulate=True)
            dx = dx.view(xshape)
        sinks = [_sink(t) for t in ctx.refs]
        if all(k is not None for k in sinks):
            with _Side(dy2, x2):
                if adjacent and _adjacent(sinks[0], sinks[2]) and _adjacent(sinks[1], sinks[3]):      # the gradients mirror the layout
                    ops.gemm(dy2, x2, torch.as_strided(sinks[0], (NT, K), (K, 1)), NT, K, M, a_mode=1, b_mode=1, lda=NT, accumulate=True,
                             split_k=ops.pick_split_k(NT, K, M), colsum_out=torch.as_strided(sinks[1], (NT,), (1,)))
                else:
                    ops.gemm(dy2, x2, sinks[0], N1, K, M, a_mode=1, b_mode=1, lda=NT, accumulate=True, split_k=ops.pick_split_k(N1, K, M),
                             colsum_out=sinks[1])
                    ops.gemm(dy2[:, N1:], x2, sinks[2], N2, K, M, a_mode=1, b_mode=1, lda=NT, accumulate=True,
                             split_k=ops.pick_split_k(N2, K, M), colsum_out=sinks[3])
            Runtime.notify(*[_param_of(t) for t in ctx.refs])
            return dx, None, None, None, None
        dw1 = torch.zeros(N1, K, dtype=torch.float32, device=dev)
        dw2 = torch.zeros(N2, K, dtype=torch.float32, device=dev)
        ops.gemm(dy2, x2, dw1, N1, K, M, a_mode=1, b_mode=1, lda=NT, accumulate=True, split_k=ops.pick_split_k(N1, K, M))
        ops.gemm(dy2[:, N1:], x2, dw2, N2, K, M, a_mode=1, b_mode=1, lda=NT, accumulate=True,
                 split_k=ops.pick_split_k(N2, K, M))
        db = torch.zeros(NT, dtype=torch.float32, device=dev)
        ops.colsum(dy2, M, NT, db)
        return dx, dw1, db[:N1], dw2, db[N1:]


def linear_cat2(x, w1, b1, w2, b2):
    return LinearCat2Fn.apply(x, w1, b1, w2, b2)


# ------------------------------------------------------------------------------------------------
# Convolution (NHWC, channels_last weights) + folded FrozenBN / bias + ReLU + residual
# ------------------------------------------------------------------------------------------------
def _w_phys(weight):
    """(O, C, KH, KW) channels_last parameter -> its physical (O, KH, KW, C) contiguous view."""
    wp = weight.permute(0, 2, 3, 1)
    assert wp.is_contiguous(), "conv weights must be stored channels_last"
    return wp


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, stride, pad, relu, allow_split=False):
        # x (N, H, W, C) contiguous NHWC
        x = _c(x)
        N, H, W, C = x.shape
        O, C2, KH, KW = weight.shape
        assert C2 == C
        wp = _w_phys(weight)
        OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        M, K = N * OH * OW, KH * KW * C
        y = torch.empty(N, OH, OW, O, dtype=torch.float32, device=x.device)
        res = _c(residual) if residual is not None else None
        geom = (N, H, W, C, KH, KW, stride, pad, OH, OW, O)
        dense = KH == 1 and KW == 1 and stride == 1 and pad == 0
        # few output tiles over a deep contraction (the 3x3/s2 input_proj conv on C5: 512 x 256 x 18432 = 32 tiles; the 3x3
        # convolutions of layer4: 2048 x 512 x 4608 = 256 tiles of 144 k-tiles each): split K over blocks (atomic partial sums,
        # the bias rides with the first split) and apply FrozenBN / ReLU / shortcut in a separate in-place pass.  Training
        # only: the atomic k-split sums in arrival order, and inference keeps run-to-run bitwise reproducibility (the
        # replayed decode graphs are tested bit-for-bit against the eager loop).
        # (`allow_split` = grad mode at the call site: autograd runs Function.forward itself with grad mode off)
        sk = ops.pick_split_k(M, O, K) if (allow_split and not _DETERMINISTIC) else 1
        plain = scale is None and res is None and not relu
        sk = sk if (sk >= 8 or (sk >= 4 and not plain)) else 1
        if sk > 1:
            y.zero_()
            kw = dict(bias=shift if plain else None, split_k=sk, accumulate=True)
        else:
            kw = dict(scale=scale, bias=shift, residual=res, relu=relu)
        if dense:
            ops.gemm(x, wp, y, M, O, K, **kw)
        else:
            ops.gemm(x, wp, y, M, O, K, a_mode=2, b_mode=0, conv=geom, **kw)
        if sk > 1 and not plain:
            ops.affine_act_(y, scale, shift, res, relu)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.w_ref, ctx.shift_ref = weight, shift
        ctx.meta = (geom, dense, relu, shift is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, scale, y = ctx.saved_tensors
        geom, dense, relu, has_shift, has_res = ctx.meta
        N, H, W, C, KH, KW, stride, pad, OH, OW, O = geom
        dy = _c(dy)
        M, K = N * OH * OW, KH * KW * C
        want_res = has_res and ctx.needs_input_grad[4]
        if relu or scale is not None or want_res:
            dpre, dres = ops.bn_relu_bwd(dy, y if relu else dy, scale, relu, want_res) if scale is not None else \
                _relu_bwd_noscale(dy, y, relu, want_res)
        else:
            dpre, dres = dy, None
        wp = _w_phys(weight)
        dx = dw = dshift = None
        if ctx.needs_input_grad[0]:
            # `accum_dx` (set by BottleneckFn): a gradient of the same input that already exists (the shortcut branch's) --
            # the data gradient is added into it by the accumulate epilogue instead of a separate summation pass
            acc_dx = getattr(ctx, "accum_dx", None)
            dx = acc_dx if acc_dx is not None else torch.empty(N, H, W, C, dtype=torch.float32, device=dy.device)
            if dense:
                ops.gemm(dpre, wp, dx, M, C, O, a_mode=0, b_mode=1, accumulate=acc_dx is not None)
            elif _dgrad_stride2_ok(geom):
                _dgrad_stride2(dpre, wp, geom, dx, acc_dx)
            else:
                sk = ops.pick_split_k(N * H * W, C, KH * KW * O)
                sk = sk if sk >= 4 else 1
                if sk > 1 and acc_dx is None:
                    dx.zero_()
                ops.gemm(dpre, wp, dx, N * H * W, C, KH * KW * O, a_mode=3, b_mode=2, conv=geom, split_k=sk,
                         accumulate=sk > 1 or acc_dx is not None)
        wsink = _sink(ctx.w_ref)
        ssink = _sink(ctx.shift_ref) if (has_shift and ctx.needs_input_grad[3]) else None
        need_w = ctx.needs_input_grad[1]
        need_s = has_shift and ctx.needs_input_grad[3]
        src_s = None
        if need_s:
            src_s = dpre if scale is None else (dres if dres is not None else _mask_only(dy, y, relu))

        def wgrad(dst_phys, bias_sums=None):
            if dense:
                ops.gemm(dpre, x, dst_phys, O, C, M, a_mode=1, b_mode=1, lda=O, ldb=C, accumulate=True,
                         split_k=ops.pick_split_k(O, C, M), colsum_out=bias_sums)
            else:
                ops.gemm(dpre, x, dst_phys, O, K, M, a_mode=1, b_mode=3, lda=O, conv=geom, accumulate=True,
                         split_k=ops.pick_split_k(O, K, M), colsum_out=bias_sums)

        if (need_w and wsink is not None) or (need_s and ssink is not None):
            with _Side(dpre, x, src_s):
                fuse_s = need_s and ssink is not None and need_w and wsink is not None and src_s is dpre
                if need_w and wsink is not None:
                    wgrad(_w_phys(wsink), ssink if fuse_s else None)
                if need_s and ssink is not None and not fuse_s:
                    ops.colsum(src_s, M, O, ssink)
            Runtime.notify(_param_of(ctx.w_ref) if (need_w and wsink is not None) else None,
                           _param_of(ctx.shift_ref) if (need_s and ssink is not None) else None)
        if need_w and wsink is None:
            dwp = torch.zeros(O, KH, KW, C, dtype=torch.float32, device=dy.device)
            wgrad(dwp)
            dw = dwp.permute(0, 3, 1, 2)
        if need_s and ssink is None:
            dshift = torch.zeros(O, dtype=torch.float32, device=dy.device)
            ops.colsum(src_s, M, O, dshift)
        return dx, dw, None, dshift, dres, None, None, None, None


_DGRAD_S2 = os.environ.get("CAPE_DGRAD_S2_CLASSES", "1") == "1"


def _dgrad_stride2_ok(geom):
    N, H, W, C, KH, KW, stride, pad, OH, OW, O = geom
    return (_DGRAD_S2 and stride == 2 and H % 2 == 0 and W % 2 == 0 and OH * 2 == H and OW * 2 == W and O % 32 == 0 and C % 4 == 0
            and (KH, KW, pad) in ((3, 3, 1), (1, 1, 0)))


def _dgrad_stride2(dpre, wp, geom, dx, acc_dx):
    """Data gradient of a stride-2 convolution (3x3 / pad 1, or 1x1) by input-parity classes.  As one gather launch over the
    full-resolution grid 3 of 4 (pixel, tap) pairs multiply structural zeros (a tap reaches a pixel only when the parities fit).
    The pixels (2y' + py, 2x' + px) of one class see a fixed subset of the taps and their gradient is a STRIDE-1 data gradient on
    the half-resolution grid: rows py = 1 use taps kh in {0, 2} with padding 1, rows py = 0 the tap kh = 1 with padding 0 (same
    for columns): 4 + 2 + 2 + 1 = 9 taps over 4 pixels instead of 36.  The four compact results go back onto the grid (and onto a
    gradient that already exists there: the shortcut branch's) in one pass.  1x1 / stride 2: only the (even, even) class is
    non-zero and it is a dense product."""
    N, H, W, C, KH, KW, stride, pad, OH, OW, O = geom
    H2, W2 = H // 2, W // 2
    Mc = N * H2 * W2
    dev = dpre.device
    classes = [None] * 4
    if KH == 1:
        c00 = torch.empty(N, H2, W2, C, dtype=torch.float32, device=dev)
        ops.gemm(dpre, wp, c00, Mc, C, O, a_mode=0, b_mode=1)
        classes[0] = c00
    else:
        for py in (0, 1):
            for px in (0, 1):
                kh_n, kw_n = (2 if py else 1), (2 if px else 1)
                sub_geom = (N, H2, W2, C, kh_n, kw_n, 1, (1 if py else 0), OH, OW, O)
                sub = ((1 if px else 0), KH, KW, (0 if py else 1), 2, (0 if px else 1), 2)
                K = kh_n * kw_n * O
                buf = torch.empty(N, H2, W2, C, dtype=torch.float32, device=dev)
                sk = ops.pick_split_k(Mc, C, K)
                sk = sk if sk >= 4 else 1
                if sk > 1:
                    buf.zero_()
                ops.gemm(dpre, wp, buf, Mc, C, K, a_mode=3, b_mode=2, conv=sub_geom, conv_sub=sub, split_k=sk, accumulate=sk > 1)
                classes[py * 2 + px] = buf
    ops.interleave2x2(classes, acc_dx, dx)


def _relu_bwd_noscale(dy, y, relu, want_res):
    if not relu:
        return dy, (dy if want_res else None)
    d = ops.relu_drop_bwd(dy, y, 1.0)
    return d, (d if want_res else None)


def _mask_only(dy, y, relu):
    return ops.relu_drop_bwd(dy, y, 1.0) if relu else dy


class _PlainCtx:
    """Stand-in for an autograd context when one Function runs several ConvFn stages itself."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()
        self.accum_dx = None

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class BottleneckFn(torch.autograd.Function):
    """torchvision Bottleneck v1.5 with FrozenBatchNorm2d (reference backbone.py:20-57 over torchvision.models.resnet50) as ONE
    autograd node: conv1-bn-relu, conv2(3x3, stride)-bn-relu, [downsample conv-bn], conv3-bn + shortcut + relu.  The backward
    runs the four ConvFn stages in order and lets conv1's data gradient accumulate into the shortcut's gradient (the block
    input has two consumers; as separate nodes their gradients cost a summation pass over the largest tensors of the trunk),
    and the host pays for one node instead of five."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, s1, b1, s2, b2, s3, b3, sd, bd, stride, allow_split):
        nx = ctx.needs_input_grad[0]
        c1 = _PlainCtx((nx, w1.requires_grad, False, False, False))
        o1 = ConvFn.forward(c1, x, w1, s1, b1, None, 1, 0, True, allow_split)
        c2 = _PlainCtx((True, w2.requires_grad, False, False, False))
        o2 = ConvFn.forward(c2, o1, w2, s2, b2, None, stride, 1, True, allow_split)
        cd, idt = None, x
        if wd is not None:
            cd = _PlainCtx((nx, wd.requires_grad, False, False, False))
            idt = ConvFn.forward(cd, x, wd, sd, bd, None, stride, 0, False, allow_split)
        c3 = _PlainCtx((True, w3.requires_grad, False, False, nx or wd is not None))
        y = ConvFn.forward(c3, o2, w3, s3, b3, idt, 1, 0, True, allow_split)
        # every tensor the four stages keep goes through the node's own save_for_backward: the block output `y` is among them
        # (conv3's ReLU mask), and an output held as a plain attribute of the context is a reference cycle through its grad_fn
        # that Python's collector cannot see -- round 2 leaked ~1.4 GiB of trunk activations per training step that way
        flat, spans = [], []
        for c in (c1, c2, c3, cd):
            if c is None:
                spans.append(None)
                continue
            spans.append((len(flat), len(c.saved_tensors)))
            flat.extend(c.saved_tensors)
            c.saved_tensors = ()
        ctx.save_for_backward(*flat)
        ctx.sub = (c1, c2, c3, cd)
        ctx.spans = spans
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, c2, c3, cd = ctx.sub
        saved = ctx.saved_tensors
        for c, sp in zip(ctx.sub, ctx.spans):
            if c is not None:
                c.saved_tensors = tuple(saved[sp[0]:sp[0] + sp[1]])
        try:
            return BottleneckFn._backward(ctx, dy)
        finally:
            for c in ctx.sub:                           # nothing of this pass stays on the long-lived stage contexts
                if c is not None:
                    c.saved_tensors, c.accum_dx = (), None

    @staticmethod
    def _backward(ctx, dy):
        c1, c2, c3, cd = ctx.sub
        d2, dw3, _, _, dres = ConvFn.backward(c3, dy)[:5]
        d1, dw2 = ConvFn.backward(c2, d2)[:2]
        dwd = None
        if cd is not None:
            dxd, dwd = ConvFn.backward(cd, dres)[:2]
            c1.accum_dx = dxd                       # None when the block input needs no gradient
        else:
            c1.accum_dx = dres if c1.needs_input_grad[0] else None
        dx, dw1 = ConvFn.backward(c1, d1)[:2]
        return (dx, dw1, dw2, dw3, dwd) + (None,) * 10


def bottleneck(x, w1, w2, w3, wd, bn1, bn2, bn3, bnd, stride):
    """bnK = (scale, shift) of the folded FrozenBatchNorm2d; wd / bnd = None without a projection shortcut."""
    sd, bd = bnd if bnd is not None else (None, None)
    return BottleneckFn.apply(x, w1, w2, w3, wd, bn1[0], bn1[1], bn2[0], bn2[1], bn3[0], bn3[1], sd, bd, int(stride), torch.is_grad_enabled())


def conv_bn_act(x, weight, scale, shift, stride=1, pad=0, relu=False, residual=None):
    return ConvFn.apply(x, weight, scale, shift, residual, int(stride), int(pad), bool(relu), torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# LayerNorm(x + dropout(y)) [+ pos]
# ------------------------------------------------------------------------------------------------
class AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, pos, dropout_p, rng_stream):
        x, y, pos = _c(x), _c(y), _c(pos)
        rng = Runtime.get_rng(x.device) if dropout_p > 0 else None
        out, mean, rstd, out_pos = ops.add_layernorm_fwd(x, y, gamma, beta, pos=pos, dropout_p=dropout_p, rng=rng,
                                                         rng_stream=rng_stream)
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        ctx.refs = (gamma, beta)
        ctx.slot = _slot_of(x)
        ctx.meta = (dropout_p, rng_stream, pos is not None)
        ctx.set_materialize_grads(False)        # an unused output arrives as None, not as a zero-filled tensor
        if pos is None:
            return out
        return out, out_pos

    @staticmethod
    def backward(ctx, d_out, d_out_pos=None):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        p, stream, has_pos = ctx.meta
        g_pos = d_out_pos                       # out_pos = out + pos: its gradient reaches `pos` unchanged
        if d_out is None and d_out_pos is None:
            return None, None, None, None, None, None, None
        if d_out is None:
            d_out, d_out_pos = d_out_pos, None
        d_out, d_out_pos = _c(d_out), _c(d_out_pos)
        C = x.shape[-1]
        gs, bs = _sink(ctx.refs[0]), _sink(ctx.refs[1])
        direct = gs is not None and bs is not None
        dg = gs if direct else torch.zeros(C, dtype=torch.float32, device=x.device)
        db = bs if direct else torch.zeros(C, dtype=torch.float32, device=x.device)
        rng = Runtime.get_rng(x.device) if p > 0 else None
        dx, dy = ops.add_layernorm_bwd(d_out, d_out_pos, x, y, gamma, mean, rstd, dg, db, dropout_p=p, rng=rng,
                                       rng_stream=stream)
        if y is None or dy is not dx:               # (without dropout the kernel hands ONE buffer to x and y: not exclusively x's)
            _slot_offer(ctx.slot, dx)
        if direct:
            Runtime.notify(_param_of(ctx.refs[0]), _param_of(ctx.refs[1]))
            dg = db = None
        dpos = g_pos if (has_pos and ctx.needs_input_grad[4]) else None
        return dx, (dy if y is not None else None), dg, db, dpos, None, None


def add_layernorm(x, y, gamma, beta, pos=None, dropout_p=0.0, rng_stream=0):
    return AddLayerNormFn.apply(x, y, gamma, beta, pos, float(dropout_p), int(rng_stream))


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(_c(a), _c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AddFn.apply(a, b)


class FanOutFn(torch.autograd.Function):
    """x -> k aliases of x for k consumers.  Autograd would sum the k incoming gradients of a multiply-used tensor with k-1
    `at::add` launches; here they arrive as separate arguments and are summed by ONE pass of cape_add_n_f32 (k reads, one
    write).  Gradients that autograd reports as None (an unused alias) are skipped."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs, seen = [], set()
        for g in grads:                                      # consumers that accumulated into the slot's buffer report it more than once
            if g is not None and (g.data_ptr(), g.numel()) not in seen:
                seen.add((g.data_ptr(), g.numel()))
                gs.append(g)
        if not gs:
            return None, None
        if len(gs) > 1 and len(gs) <= 8 and not all(g.is_contiguous() for g in gs) and all(_row_strided(g) for g in gs):
            return ops.add_n_rows(gs), None                  # a summand is a column block of a wider buffer: summed where it lies
        gs = [_c(g) for g in gs]
        out = gs[0]
        for i in range(0, len(gs) - 1, 7):                  # 8 sources per launch
            out = ops.add_n([out] + gs[1 + i:8 + i]) if len(gs) > 1 else out
        return out, None


def _row_strided(g):
    if g.dim() < 2 or g.stride(-1) != 1 or g.shape[-1] % 4 or g.data_ptr() % 16 or g.stride(-2) % 4:
        return False
    st, sh = g.stride(), g.shape
    return all(st[i] == st[i + 1] * sh[i + 1] for i in range(len(sh) - 2))


def fanout(x, k):
    """k aliases of x whose gradients are summed by one HIP launch (no-op outside autograd or for k == 1)."""
    if k == 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * k
    outs = FanOutFn.apply(x, k)
    slot = _Slot()
    for o in outs:
        o._cape_slot = slot                                  # consumers find the shared gradient slot on their input (see _Slot)
    return outs


# ------------------------------------------------------------------------------------------------
# pieces of the bidirectional cross-attention blocks (models/bixattn.py): exact GELU, LayerScale residual, attention cores over
# [r | v] projections.  Dropout-free (the reference never trains these blocks; the fixtures are taken with every rate at 0)
# ------------------------------------------------------------------------------------------------
class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, _c(g))


def gelu(x):
    return GeluFn.apply(x)


class ScaleResidualFn(torch.autograd.Function):
    """out = x + gamma * y (gamma per channel, or None = 1; x None = 0)."""

    @staticmethod
    def forward(ctx, x, y, gamma):
        y = _c(y)
        ctx.has_x = x is not None
        ctx.save_for_backward(y, gamma)
        ctx.gref = gamma
        return ops.scale_residual(_c(x) if x is not None else torch.zeros_like(y), y, gamma)

    @staticmethod
    def backward(ctx, g):
        y, gamma = ctx.saved_tensors
        g = _c(g)
        dx = g if ctx.has_x else None
        if gamma is None:
            return dx, g, None
        sink = _sink(ctx.gref)
        dgamma = sink if sink is not None else torch.zeros_like(gamma)
        dy = ops.scale_residual_bwd(g, y, gamma, dgamma)
        if sink is not None:
            Runtime.notify(_param_of(ctx.gref))
            dgamma = None
        return dx, dy, dgamma


def scale_residual(x, y, gamma=None):
    return ScaleResidualFn.apply(x, y, gamma)


class BiAttnCoreFn(torch.autograd.Function):
    """Both directions of bixattn.py:52-88 over rv_l = [r_l | v_l] (B, Nl, 2D) and rv_p = [r_p | v_p] (B, Np, 2D):
        lat = softmax_p(scale r_l r_p^T) v_p      pat = softmax_l(scale r_p r_l^T) v_l
    Backward: each direction is one pass of the attention backward kernels writing into column blocks of d_rv_l / d_rv_p; the two
    contributions to d r_l (query of one direction, key of the other) and to d r_p are added in place."""

    @staticmethod
    def forward(ctx, rv_l, rv_p, nheads, scale):
        rv_l, rv_p = _c(rv_l), _c(rv_p)
        B, Nl, D2 = rv_l.shape
        Np, D = rv_p.shape[1], D2 // 2
        lat, lse1 = ops.attn_fwd(rv_l[..., :D], rv_p[..., :D], rv_p[..., D:], B, nheads, Nl, Np, scale)
        pat, lse2 = ops.attn_fwd(rv_p[..., :D], rv_l[..., :D], rv_l[..., D:], B, nheads, Np, Nl, scale)
        ctx.save_for_backward(rv_l, rv_p, lat, pat, lse1, lse2)
        ctx.meta = (nheads, scale)
        return lat, pat

    @staticmethod
    def backward(ctx, d_lat, d_pat):
        rv_l, rv_p, lat, pat, lse1, lse2 = ctx.saved_tensors
        nheads, scale = ctx.meta
        B, Nl, D2 = rv_l.shape
        Np, D = rv_p.shape[1], D2 // 2
        d_l, d_p = torch.empty_like(rv_l), torch.empty_like(rv_p)
        t_l, t_p = torch.empty_like(rv_l), torch.empty_like(rv_p)     # (only the r halves are used: same row stride as the inputs)
        ops.attn_bwd(_c(d_lat), rv_l[..., :D], rv_p[..., :D], rv_p[..., D:], lat, lse1, d_l[..., :D], d_p[..., :D], d_p[..., D:],
                     B, nheads, Nl, Np, scale)
        ops.attn_bwd(_c(d_pat), rv_p[..., :D], rv_l[..., :D], rv_l[..., D:], pat, lse2, t_p[..., :D], t_l[..., :D], d_l[..., D:],
                     B, nheads, Np, Nl, scale)
        ops.add_n_rows([d_l[..., :D], t_l[..., :D]], out=d_l[..., :D])
        ops.add_n_rows([d_p[..., :D], t_p[..., :D]], out=d_p[..., :D])
        return d_l, d_p, None, None


def bi_attn_core(rv_l, rv_p, nheads, scale):
    return BiAttnCoreFn.apply(rv_l, rv_p, nheads, float(scale))


class AttnKVFn(torch.autograd.Function):
    """One direction (bixattn.py:90-119): out = softmax(scale r_q r_k^T) v_k with [r_k | v_k] = rv_kv (B, Nk, 2D)."""

    @staticmethod
    def forward(ctx, r_q, rv_kv, nheads, scale):
        r_q, rv_kv = _c(r_q), _c(rv_kv)
        B, Nq, D = r_q.shape
        Nk = rv_kv.shape[1]
        out, lse = ops.attn_fwd(r_q, rv_kv[..., :D], rv_kv[..., D:], B, nheads, Nq, Nk, scale)
        ctx.save_for_backward(r_q, rv_kv, out, lse)
        ctx.meta = (nheads, scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        r_q, rv_kv, out, lse = ctx.saved_tensors
        nheads, scale = ctx.meta
        B, Nq, D = r_q.shape
        Nk = rv_kv.shape[1]
        dq, dkv = torch.empty_like(r_q), torch.empty_like(rv_kv)
        ops.attn_bwd(_c(d_out), r_q, rv_kv[..., :D], rv_kv[..., D:], out, lse, dq, dkv[..., :D], dkv[..., D:], B, nheads, Nq, Nk, scale)
        return dq, dkv, None, None


def attn_kv(r_q, rv_kv, nheads, scale):
    return AttnKVFn.apply(r_q, rv_kv, nheads, float(scale))


# ------------------------------------------------------------------------------------------------
# input_proj GroupNorm of all levels, written straight into the flattened token buffer
# ------------------------------------------------------------------------------------------------
class LevelGroupNormFn(torch.autograd.Function):
    """xs: L tensors (N, h_l, w_l, C) NHWC; returns src_flatten (N, S, C)."""

    @staticmethod
    def forward(ctx, geo, *args):
        L = geo.L
        xs, gammas, betas = args[:L], args[L:2 * L], args[2 * L:3 * L]
        N, C = xs[0].shape[0], xs[0].shape[-1]
        out = torch.empty(N, geo.S, C, dtype=torch.float32, device=xs[0].device)
        stats = []
        xs = [_c(x) for x in xs]
        for l, x in enumerate(xs):
            h, w = geo.shapes[l]
            assert x.shape[1] == h and x.shape[2] == w
            stats.append(ops.groupnorm_fwd(x, gammas[l], betas[l], out[:, geo.starts[l]:], geo.S * C, N, h * w, C))
        ctx.geo = geo
        ctx.refs = (tuple(gammas), tuple(betas))
        ctx.save_for_backward(*xs, *gammas, *[s for st in stats for s in st])
        return out

    @staticmethod
    def backward(ctx, d_out):
        geo = ctx.geo
        L = geo.L
        sv = ctx.saved_tensors
        xs, gammas, st = sv[:L], sv[L:2 * L], sv[2 * L:]
        d_out = _c(d_out)
        N, S, C = d_out.shape
        dxs, dgs, dbs = [], [], []
        for l in range(L):
            h, w = geo.shapes[l]
            gs, bs = _sink(ctx.refs[0][l]), _sink(ctx.refs[1][l])
            direct = gs is not None and bs is not None
            dg = gs if direct else torch.zeros(C, dtype=torch.float32, device=d_out.device)
            db = bs if direct else torch.zeros(C, dtype=torch.float32, device=d_out.device)
            dx = ops.groupnorm_bwd(d_out[:, geo.starts[l]:], S * C, xs[l], gammas[l], st[2 * l], st[2 * l + 1], dg, db, N,
                                   h * w, C)
            if direct:
                Runtime.notify(_param_of(ctx.refs[0][l]), _param_of(ctx.refs[1][l]))
                dg = db = None
            dxs.append(dx); dgs.append(dg); dbs.append(db)
        return (None, *dxs, *dgs, *dbs)


def level_groupnorm(geo, xs, gammas, betas):
    return LevelGroupNormFn.apply(geo, *xs, *gammas, *betas)


_SINE_CACHE = {}


class LevelPosFn(torch.autograd.Function):
    """Image sine position embedding + level_embed, flattened (N, S, C)."""

    @staticmethod
    def forward(ctx, geo, level_embed, *masks_u8):
        N = masks_u8[0].shape[0]
        C = level_embed.shape[1]
        out = torch.empty(N, geo.S, C, dtype=torch.float32, device=level_embed.device)
        # unpadded batches hand in the cached all-False masks (util/misc.cached_zero_mask): the sine part is then a constant of the
        # geometry, kept once (44.5 MB at 32 x 1360 x 256) -- a step only adds the trainable level_embed rows to it
        key = (tuple(m.data_ptr() for m in masks_u8), tuple(geo.shapes), N, C, str(level_embed.device))
        sine = _SINE_CACHE.get(key) if all(getattr(m, "_cape_all_false", False) for m in masks_u8) else None
        if sine is None:
            target = out
            if all(getattr(m, "_cape_all_false", False) for m in masks_u8) and not capturing():
                target = torch.empty_like(out)
            zero_row = torch.zeros(C, dtype=torch.float32, device=level_embed.device) if target is not out else None
            for l, m in enumerate(masks_u8):
                h, w = geo.shapes[l]
                ops.pos_sine_level(_c(m), zero_row if zero_row is not None else level_embed[l], target[:, geo.starts[l]:], geo.S * C, N, h, w, C)
            if target is not out:
                if len(_SINE_CACHE) >= 8:
                    _SINE_CACHE.pop(next(iter(_SINE_CACHE)))
                _SINE_CACHE[key] = sine = target
        if sine is not None:
            ops.level_embed_add(sine, _c(level_embed), geo, out)
        ctx.geo = geo
        ctx.N = N
        ctx.le_ref = level_embed
        return out

    @staticmethod
    def backward(ctx, d_pos):
        geo, N = ctx.geo, ctx.N
        d_pos = _c(d_pos)
        C = d_pos.shape[-1]
        sink = _sink(ctx.le_ref)
        d_le = sink if sink is not None else torch.zeros(geo.L, C, dtype=torch.float32, device=d_pos.device)
        for l in range(geo.L):
            h, w = geo.shapes[l]
            ops.colsum(d_pos[0, geo.starts[l]:], h * w, C, d_le[l], nbatch=N, batch_stride=geo.S * C)
        if sink is not None:
            Runtime.notify(_param_of(ctx.le_ref))
            d_le = None
        return (None, d_le) + (None,) * geo.L


def level_pos(geo, level_embed, masks_u8):
    return LevelPosFn.apply(geo, level_embed, *masks_u8)


# ------------------------------------------------------------------------------------------------
# MSDA core
# ------------------------------------------------------------------------------------------------
class MSDAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, offw, ref, geo, P):
        value, offw, ref = _c(value), _c(offw), _c(ref)
        N, Lq = offw.shape[0], offw.shape[1]
        out = ops.msda_fwd(value, offw, ref, geo, N, Lq, P)
        ctx.save_for_backward(value, offw, ref)
        ctx.meta = (geo, N, Lq, P)
        return out

    @staticmethod
    def backward(ctx, d_out):
        value, offw, ref = ctx.saved_tensors
        geo, N, Lq, P = ctx.meta
        dv, do, dr = ops.msda_bwd(_c(d_out), value, offw, ref, geo, N, Lq, P, need_ref_grad=ctx.needs_input_grad[2])
        return dv, do, dr, None, None


def msda(value, offw, ref, geo, P=4):
    return MSDAFn.apply(value, offw, ref, geo, P)


# ------------------------------------------------------------------------------------------------
# nn.MultiheadAttention (in_proj + core + out_proj) as one node
# ------------------------------------------------------------------------------------------------
class MHAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_in, k_in, v_in, in_w, in_b, out_w, out_b, nheads, mask_mode, kpm_u8, dropout_p, rng_stream):
        q_in, k_in, v_in = _c(q_in), _c(k_in), _c(v_in)
        N, Lq, C = q_in.shape
        Lk = k_in.shape[1]
        dev = q_in.device
        scale = (C // nheads) ** -0.5
        rng = Runtime.get_rng(dev) if dropout_p > 0 else None
        mm = ops.attn_mm_ok(N, nheads, Lq, Lk)      # long rows: contractions on the matrix cores, softmax kernel in between
        # projections of the same input are ONE launch over the stacked in_proj rows (the support encoder's self-attention:
        # q = k = v -> N = 768; cross-attention onto the support features: k = v -> N = 512); q / k / v are then column views of
        # the wide result (row stride 3C / 2C), which the short-row attention kernels take as they are
        merged = 0
        if not mm and k_in is v_in:
            merged = 3 if (q_in is k_in) else 2
        if merged == 3:
            qkv = torch.empty(N, Lq, 3 * C, dtype=torch.float32, device=dev)
            ops.gemm(q_in.view(-1, C), in_w, qkv, N * Lq, 3 * C, C, bias=in_b)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        else:
            q = torch.empty(N, Lq, C, dtype=torch.float32, device=dev)
            ops.gemm(q_in.view(-1, C), in_w, q, N * Lq, C, C, bias=in_b)
            if merged == 2:
                kv = torch.empty(N, Lk, 2 * C, dtype=torch.float32, device=dev)
                ops.gemm(k_in.view(-1, C), in_w[C:], kv, N * Lk, 2 * C, C, bias=in_b[C:])
                k, v = kv[..., :C], kv[..., C:]
            else:
                k = torch.empty(N, Lk, C, dtype=torch.float32, device=dev)
                v = torch.empty(N, Lk, C, dtype=torch.float32, device=dev)
                ops.gemm(k_in.view(-1, C), in_w[C:], k, N * Lk, C, C, bias=in_b[C:])
                ops.gemm(v_in.view(-1, C), in_w[2 * C:], v, N * Lk, C, C, bias=in_b[2 * C:])
        if mm:
            O, Pp, Pu = ops.attn_mm_fwd(q, k, v, N, nheads, Lq, Lk, scale, mask_mode=mask_mode, kpm=kpm_u8, dropout_p=dropout_p,
                                        rng=rng, rng_stream=rng_stream)
            lse = Pp
            ctx.Pu = Pu
        else:
            O, lse = ops.attn_fwd(q, k, v, N, nheads, Lq, Lk, scale, mask_mode=mask_mode, kpm=kpm_u8, dropout_p=dropout_p,
                                  rng=rng, rng_stream=rng_stream)
        out = torch.empty(N, Lq, C, dtype=torch.float32, device=dev)
        ops.gemm(O.view(-1, C), out_w, out, N * Lq, C, C, bias=out_b)
        ctx.save_for_backward(q_in, k_in, v_in, in_w, out_w, q, k, v, O, lse, kpm_u8)
        ctx.refs = (in_w, in_b, out_w, out_b)
        ctx.slot_q = _slot_of(q_in)
        ctx.meta = (nheads, mask_mode, dropout_p, rng_stream, scale, k_in is v_in, q_in is k_in, mm, merged)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q_in, k_in, v_in, in_w, out_w, q, k, v, O, lse, kpm = ctx.saved_tensors
        nheads, mask_mode, p, stream, scale, kv_same, qk_same, mm, merged = ctx.meta
        N, Lq, C = q_in.shape
        Lk = k_in.shape[1]
        dev = d_out.device
        d_out2 = _c(d_out).view(-1, C)
        Mq, Mk = N * Lq, N * Lk
        dO = torch.empty(Mq, C, dtype=torch.float32, device=dev)
        ops.gemm(d_out2, out_w, dO, Mq, C, C, a_mode=0, b_mode=1)
        sinks = [_sink(t) for t in ctx.refs]
        direct = all(k is not None for k in sinks)
        d_out_w = sinks[2] if direct else torch.zeros(C, C, dtype=torch.float32, device=dev)
        d_out_b = sinks[3] if direct else torch.zeros(C, dtype=torch.float32, device=dev)
        d_in_w = sinks[0] if direct else torch.zeros(3 * C, C, dtype=torch.float32, device=dev)
        d_in_b = sinks[1] if direct else torch.zeros(3 * C, dtype=torch.float32, device=dev)

        def out_grads():
            ops.gemm(d_out2, O.view(-1, C), d_out_w, C, C, Mq, a_mode=1, b_mode=1, accumulate=True,
                     split_k=ops.pick_split_k(C, C, Mq), colsum_out=d_out_b)

        if direct:
            with _Side(d_out2, O):
                out_grads()
        else:
            out_grads()
        if merged == 3:                              # gradients in the layout of the wide projection result
            dqkv = torch.empty(N, Lq, 3 * C, dtype=torch.float32, device=dev)
            dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        elif merged == 2:
            dq = torch.empty(N, Lq, C, dtype=torch.float32, device=dev)
            dkv = torch.empty(N, Lk, 2 * C, dtype=torch.float32, device=dev)
            dk, dv = dkv[..., :C], dkv[..., C:]
        else:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        rng = Runtime.get_rng(dev) if p > 0 else None
        if mm:
            ops.attn_mm_bwd(dO.view(N, Lq, C), q, k, v, lse, ctx.Pu, dq, dk, dv, N, nheads, Lq, Lk, scale, dropout_p=p, rng=rng,
                            rng_stream=stream)
        else:
            ops.attn_bwd(dO.view(N, Lq, C), q, k, v, O, lse, dq, dk, dv, N, nheads, Lq, Lk, scale, mask_mode=mask_mode, kpm=kpm,
                         dropout_p=p, rng=rng, rng_stream=stream)

        def in_grads():
            if merged == 3:                          # [dq | dk | dv]^T x: one product for the stacked in_proj rows
                ops.gemm(dqkv.view(-1, 3 * C), q_in.view(-1, C), d_in_w, 3 * C, C, Mq, a_mode=1, b_mode=1, lda=3 * C, accumulate=True,
                         split_k=ops.pick_split_k(3 * C, C, Mq), colsum_out=d_in_b)
                return
            ops.gemm(dq.view(-1, C), q_in.view(-1, C), d_in_w, C, C, Mq, a_mode=1, b_mode=1, accumulate=True,
                     split_k=ops.pick_split_k(C, C, Mq), colsum_out=d_in_b)
            if merged == 2:
                ops.gemm(dkv.view(-1, 2 * C), k_in.view(-1, C), d_in_w[C:], 2 * C, C, Mk, a_mode=1, b_mode=1, lda=2 * C, accumulate=True,
                         split_k=ops.pick_split_k(2 * C, C, Mk), colsum_out=d_in_b[C:])
                return
            for i, (g, src, M) in ((1, (dk, k_in, Mk)), (2, (dv, v_in, Mk))):
                ops.gemm(g.view(-1, C), src.view(-1, C), d_in_w[i * C:], C, C, M, a_mode=1, b_mode=1, accumulate=True,
                         split_k=ops.pick_split_k(C, C, M), colsum_out=d_in_b[i * C:])

        if direct:
            with _Side(dq, dk, dv, q_in, k_in, v_in):
                in_grads()
            Runtime.notify(*[_param_of(t) for t in ctx.refs])
            d_in_w = d_in_b = d_out_w = d_out_b = None
        else:
            in_grads()
        # input gradients; when the same tensor came in as k and v (support features) or as q, k and v (self-attention of the
        # support encoder) the products accumulate into ONE buffer (GEMM epilogue C += ...) and the duplicates report None
        dq_in = dk_in = dv_in = None
        need_q, need_k, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if merged == 3 and (need_q or need_k or need_v):
            # d x = [dq | dk | dv] . in_proj_weight: one product with K = 3C; the one tensor reports through its first live slot
            dx = torch.empty(N, Lq, C, dtype=torch.float32, device=dev)
            ops.gemm(dqkv.view(-1, 3 * C), in_w, dx, Mq, C, 3 * C, a_mode=0, b_mode=1)
            return ((dx if need_q else None), (dx if (need_k and not need_q) else None), (dx if (need_v and not (need_q or need_k)) else None),
                    d_in_w, d_in_b, d_out_w, d_out_b, None, None, None, None, None)
        if merged == 2:
            if need_k or need_v:
                dk_in = torch.empty(N, Lk, C, dtype=torch.float32, device=dev)
                ops.gemm(dkv.view(-1, 2 * C), in_w[C:], dk_in, Mk, C, 2 * C, a_mode=0, b_mode=1)
                if not need_k:
                    dv_in, dk_in = dk_in, None
            if need_q:
                dq_in, acc = _grad_target(ctx.slot_q, (N, Lq, C), dev)
                ops.gemm(dq.view(-1, C), in_w, dq_in, Mq, C, C, a_mode=0, b_mode=1, accumulate=acc)
            return dq_in, dk_in, dv_in, d_in_w, d_in_b, d_out_w, d_out_b, None, None, None, None, None
        if need_k:
            dk_in = torch.empty(N, Lk, C, dtype=torch.float32, device=dev)
            ops.gemm(dk.view(-1, C), in_w[C:], dk_in, Mk, C, C, a_mode=0, b_mode=1)
        if need_v:
            if kv_same and dk_in is not None:
                ops.gemm(dv.view(-1, C), in_w[2 * C:], dk_in, Mk, C, C, a_mode=0, b_mode=1, accumulate=True)
            else:
                dv_in = torch.empty(N, Lk, C, dtype=torch.float32, device=dev)
                ops.gemm(dv.view(-1, C), in_w[2 * C:], dv_in, Mk, C, C, a_mode=0, b_mode=1)
        if need_q:
            if qk_same and dk_in is not None:
                ops.gemm(dq.view(-1, C), in_w, dk_in, Mq, C, C, a_mode=0, b_mode=1, accumulate=True)
            else:
                dq_in = torch.empty(N, Lq, C, dtype=torch.float32, device=dev)
                ops.gemm(dq.view(-1, C), in_w, dq_in, Mq, C, C, a_mode=0, b_mode=1)
        return dq_in, dk_in, dv_in, d_in_w, d_in_b, d_out_w, d_out_b, None, None, None, None, None


class DecSelfAttnFn(torch.autograd.Function):
    """The decoder layer's causal self-attention block (deformable_transformer_v2.py:323-341) as ONE node:
        q = attn_q(tgt) + query_pos ; k = attn_k(tgt) ; v = attn_v(tgt) ; nn.MultiheadAttention(q, k, v, causal mask)
    Round 2 ran it as 3 + 3 projection launches around the attention core (and 3 + 3 data-gradient launches, 7 weight-gradient
    launches and a 4-way gradient sum in the backward).  Here: the three bias-free projections are one product over the stacked
    (3C, C) weight rows with the `+ query_pos` epilogue on the q columns (attn_q / attn_k / attn_v sit back to back in the flat
    arena); the three in_proj blocks are one batch-3 launch (block i multiplies columns [iC, (i+1)C) of the first product);
    the backward mirrors it (one batch-3 data gradient, ONE K = 3C product for d tgt -- the gradient sum over the three
    consumers of `tgt` happens inside the contraction), and the seven weight gradients join the deferred groups."""

    @staticmethod
    def forward(ctx, tgt, pos, wq, wk, wv, in_w, in_b, out_w, out_b, nheads, dropout_p, rng_stream):
        N, L, C = tgt.shape
        M = N * L
        dev = tgt.device
        x, pos2 = _c(tgt).view(M, C), _c(pos).view(M, C)
        stacked = _adjacent(wq, wk) and _adjacent(wk, wv)
        qkv1 = torch.empty(M, 3 * C, dtype=torch.float32, device=dev)
        if stacked:
            ops.gemm(x, torch.as_strided(wq, (3 * C, C), (C, 1)), qkv1, M, 3 * C, C, residual=pos2, ldr=C, res_cols=C)
        else:
            ops.gemm(x, wq, qkv1, M, C, C, ldc=3 * C, residual=pos2, ldr=C)
            ops.gemm(x, wk, qkv1[:, C:], M, C, C, ldc=3 * C)
            ops.gemm(x, wv, qkv1[:, 2 * C:], M, C, C, ldc=3 * C)
        qkv2 = torch.empty(N, L, 3 * C, dtype=torch.float32, device=dev)
        ops.gemm(qkv1, in_w, qkv2, M, C, C, lda=3 * C, ldb=C, ldc=3 * C, bias=in_b,
                 batch=(3, 3, 0, C, 0, C * C, 0, C), bias_strides=(0, C))
        q, k, v = qkv2[..., :C], qkv2[..., C:2 * C], qkv2[..., 2 * C:]
        scale = (C // nheads) ** -0.5
        rng = Runtime.get_rng(dev) if dropout_p > 0 else None
        flash = ops.flash_attn_ok(N, nheads, L, L)
        mm = (not flash) and ops.attn_mm_ok(N, nheads, L, L)
        if flash:
            O, lse = ops.flash_attn_fwd(q, k, v, N, nheads, L, L, scale, mask_mode=1, dropout_p=dropout_p, rng=rng, rng_stream=rng_stream)
        elif mm:
            O, Pp, Pu = ops.attn_mm_fwd(q, k, v, N, nheads, L, L, scale, mask_mode=1, dropout_p=dropout_p, rng=rng, rng_stream=rng_stream)
            lse = Pp
            ctx.Pu = Pu
        else:
            O, lse = ops.attn_fwd(q, k, v, N, nheads, L, L, scale, mask_mode=1, dropout_p=dropout_p, rng=rng, rng_stream=rng_stream)
        out = torch.empty(N, L, C, dtype=torch.float32, device=dev)
        ops.gemm(O.view(M, C), out_w, out, M, C, C, bias=out_b)
        ctx.save_for_backward(x, wq, wk, wv, in_w, out_w, qkv1, qkv2, O, lse)
        ctx.refs = (wq, wk, wv, in_w, in_b, out_w, out_b)
        ctx.meta = (nheads, dropout_p, rng_stream, scale, mm, stacked, N, L, C)
        ctx.flash = flash
        ctx.slot = _slot_of(tgt)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, wq, wk, wv, in_w, out_w, qkv1, qkv2, O, lse = ctx.saved_tensors
        nheads, p, stream, scale, mm, stacked, N, L, C = ctx.meta
        M = N * L
        dev = d_out.device
        d_out2 = _c(d_out).view(M, C)
        sinks = [_sink(t) for t in ctx.refs]
        direct = all(k is not None for k in sinks)
        g = (lambda i, shape: sinks[i] if direct else torch.zeros(shape, dtype=torch.float32, device=dev))
        d_wq, d_wk, d_wv = g(0, (C, C)), g(1, (C, C)), g(2, (C, C))
        d_in_w, d_in_b, d_out_w, d_out_b = g(3, (3 * C, C)), g(4, (3 * C,)), g(5, (C, C)), g(6, (C,))
        dO = torch.empty(M, C, dtype=torch.float32, device=dev)
        ops.gemm(d_out2, out_w, dO, M, C, C, a_mode=0, b_mode=1)
        dqkv2 = torch.empty(N, L, 3 * C, dtype=torch.float32, device=dev)
        dq, dk, dv = dqkv2[..., :C], dqkv2[..., C:2 * C], dqkv2[..., 2 * C:]
        q, k, v = qkv2[..., :C], qkv2[..., C:2 * C], qkv2[..., 2 * C:]
        rng = Runtime.get_rng(dev) if p > 0 else None
        if ctx.flash:
            ops.flash_attn_bwd(dO.view(N, L, C), q, k, v, O, lse, dq, dk, dv, N, nheads, L, L, scale, mask_mode=1, dropout_p=p, rng=rng,
                               rng_stream=stream)
        elif mm:
            ops.attn_mm_bwd(dO.view(N, L, C), q, k, v, lse, ctx.Pu, dq, dk, dv, N, nheads, L, L, scale, dropout_p=p, rng=rng, rng_stream=stream)
        else:
            ops.attn_bwd(dO.view(N, L, C), q, k, v, O, lse, dq, dk, dv, N, nheads, L, L, scale, mask_mode=1, dropout_p=p, rng=rng,
                         rng_stream=stream)
        dqkv1 = torch.empty(M, 3 * C, dtype=torch.float32, device=dev)
        ops.gemm(dqkv2.view(M, 3 * C), in_w, dqkv1, M, C, C, a_mode=0, b_mode=1, lda=3 * C, ldb=C, ldc=3 * C,
                 batch=(3, 3, 0, C, 0, C * C, 0, C))
        need_x, need_pos = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = None
        if need_x:
            dx, acc = _grad_target(ctx.slot, (N, L, C), dev)
            if stacked:
                ops.gemm(dqkv1, torch.as_strided(wq, (3 * C, C), (C, 1)), dx, M, C, 3 * C, a_mode=0, b_mode=1, accumulate=acc)
            else:
                for i, w in enumerate((wq, wk, wv)):
                    ops.gemm(dqkv1[:, i * C:], w, dx, M, C, C, a_mode=0, b_mode=1, lda=3 * C, accumulate=acc or i > 0)

        def wgrads():
            Mv = dqkv2.view(M, 3 * C)
            ops.gemm(d_out2, O.view(M, C), d_out_w, C, C, M, a_mode=1, b_mode=1, accumulate=True, split_k=ops.pick_split_k(C, C, M),
                     colsum_out=d_out_b)
            for i, dw in enumerate((d_wq, d_wk, d_wv)):
                ops.gemm(Mv[:, i * C:], qkv1[:, i * C:], d_in_w[i * C:], C, C, M, a_mode=1, b_mode=1, lda=3 * C, ldb=3 * C, accumulate=True,
                         split_k=ops.pick_split_k(C, C, M), colsum_out=d_in_b[i * C:])
                ops.gemm(dqkv1[:, i * C:], x, dw, C, C, M, a_mode=1, b_mode=1, lda=3 * C, ldb=C, accumulate=True,
                         split_k=ops.pick_split_k(C, C, M))

        if direct:
            with _Side(d_out2, O, dqkv2, qkv1, dqkv1, x):
                wgrads()
            Runtime.notify(*[_param_of(t) for t in ctx.refs])
            grads = (None,) * 7
        else:
            wgrads()
            grads = (d_wq, d_wk, d_wv, d_in_w, d_in_b, d_out_w, d_out_b)
        dpos = dqkv1[:, :C].view(N, L, C) if need_pos else None      # (a row-strided view of the wide gradient)
        return (dx, dpos) + grads + (None, None, None)


def dec_self_attn(tgt, pos, wq, wk, wv, in_w, in_b, out_w, out_b, nheads=8, dropout_p=0.0, rng_stream=0):
    return DecSelfAttnFn.apply(tgt, pos, wq, wk, wv, in_w, in_b, out_w, out_b, nheads, float(dropout_p), int(rng_stream))


def mha(q_in, k_in, v_in, in_w, in_b, out_w, out_b, nheads=8, mask_mode=0, kpm_u8=None, dropout_p=0.0, rng_stream=0):
    return MHAFn.apply(q_in, k_in, v_in, in_w, in_b, out_w, out_b, nheads, mask_mode, kpm_u8, float(dropout_p), int(rng_stream))


# ------------------------------------------------------------------------------------------------
# decoder embedding / reference-point ops
# ------------------------------------------------------------------------------------------------
class TokenEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, pad_idx, s11, s21, s12, s22, dx1, dx2, dy1, dy2):
        shape = s11.shape
        seqs = [_c(s).view(-1) for s in (s11, s21, s12, s22)]
        deltas = [_c(d).view(-1) for d in (dx1, dx2, dy1, dy2)]
        out = ops.token_embed_fwd(table, seqs, deltas)
        ctx.save_for_backward(*seqs, *deltas)
        ctx.t_ref = table
        ctx.meta = (table.shape, pad_idx)
        return out.view(*shape, table.shape[1])

    @staticmethod
    def backward(ctx, d_out):
        sv = ctx.saved_tensors
        tshape, pad_idx = ctx.meta
        sink = _sink(ctx.t_ref)
        d_table = sink if sink is not None else torch.zeros(tshape, dtype=torch.float32, device=d_out.device)
        ops.token_embed_bwd(_c(d_out).view(-1, tshape[1]), sv[:4], sv[4:], d_table, pad_idx if pad_idx is not None else -1)
        if sink is not None:
            Runtime.notify(_param_of(ctx.t_ref))
            d_table = None
        return (d_table,) + (None,) * 9


def token_embed(table, pad_idx, s11, s21, s12, s22, dx1, dx2, dy1, dy2):
    return TokenEmbedFn.apply(table, pad_idx, s11, s21, s12, s22, dx1, dx2, dy1, dy2)


class QuerySineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref):
        ref = _c(ref)
        ctx.save_for_backward(ref)
        return ops.query_sine_fwd(ref).view(*ref.shape[:-1], 256)

    @staticmethod
    def backward(ctx, d_out):
        (ref,) = ctx.saved_tensors
        return ops.query_sine_bwd(_c(d_out).view(-1, 256), ref).view_as(ref)


def query_sine(ref):
    return QuerySineFn.apply(ref)


class RefineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, delta, ref):
        delta, ref = _c(delta), _c(ref)
        out = ops.refine_fwd(delta, ref)
        ctx.save_for_backward(out, ref)
        return out

    @staticmethod
    def backward(ctx, d_new):
        out, ref = ctx.saved_tensors
        d_delta, d_ref = ops.refine_bwd(_c(d_new), out, ref)
        return d_delta, d_ref


def refine(delta, ref):
    return RefineFn.apply(delta, ref)


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_bwd(_c(dy), y)


def sigmoid(x):
    return SigmoidFn.apply(x)


class RefScaleFn(torch.autograd.Function):
    """ref (N, Lq, 2) * valid_ratios (N, L, 2) -> (N, Lq, L, 2)"""

    @staticmethod
    def forward(ctx, ref, valid_ratios):
        ref, vr = _c(ref), _c(valid_ratios)
        N, Lq, _ = ref.shape
        L = vr.shape[1]
        ctx.save_for_backward(vr)
        ctx.meta = (N, Lq, L)
        return ops.ref_scale_fwd(ref, vr, Lq, L).view(N, Lq, L, 2)

    @staticmethod
    def backward(ctx, d):
        (vr,) = ctx.saved_tensors
        N, Lq, L = ctx.meta
        return ops.ref_scale_bwd(_c(d), vr, Lq, L).view(N, Lq, 2), None


def ref_scale(ref, valid_ratios):
    return RefScaleFn.apply(ref, valid_ratios)


# ------------------------------------------------------------------------------------------------
# support encoder pieces
# ------------------------------------------------------------------------------------------------
class SupportEmbedFn(torch.autograd.Function):
    """coords (N,P,2) -> h = relu(Linear_2->C(coords)) (N,P,C), pe = sine2d + pe1d (N,P,C; no gradient)."""

    @staticmethod
    def forward(ctx, coords, W0, b0, pe1d):
        coords = _c(coords)
        N, P, _ = coords.shape
        C = W0.shape[0]
        h, pe = ops.support_embed_fwd(coords, W0, b0, pe1d, N, P, C)
        ctx.save_for_backward(h, coords)
        ctx.refs = (W0, b0)
        ctx.meta = (N, P, C)
        ctx.mark_non_differentiable(pe)
        ctx.set_materialize_grads(False)
        return h.view(N, P, C), pe.view(N, P, C)

    @staticmethod
    def backward(ctx, d_h, _d_pe):
        if d_h is None:
            return None, None, None, None
        h, coords = ctx.saved_tensors
        N, P, C = ctx.meta
        ws, bs = _sink(ctx.refs[0]), _sink(ctx.refs[1])
        direct = ws is not None and bs is not None
        dW = ws if direct else torch.zeros(C, 2, dtype=torch.float32, device=h.device)
        db = bs if direct else torch.zeros(C, dtype=torch.float32, device=h.device)
        ops.support_embed_bwd(_c(d_h).view(-1, C), h, coords, dW, db, N, P, C)
        if direct:
            Runtime.notify(_param_of(ctx.refs[0]), _param_of(ctx.refs[1]))
            dW = db = None
        return None, dW, db, None


def support_embed(coords, W0, b0, pe1d):
    return SupportEmbedFn.apply(coords, W0, b0, pe1d)


class GCNAggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, adj):
        y = _c(y)
        N, P, C2 = y.shape
        out = ops.gcn_aggregate_fwd(y, adj, N, P, C2 // 2)
        ctx.save_for_backward(out, adj)
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, adj = ctx.saved_tensors
        N, P, C = out.shape
        return ops.gcn_aggregate_bwd(_c(d_out), out, adj, N, P, C), None


def gcn_aggregate(y, adj):
    return GCNAggFn.apply(y, adj)


class ZeroRowsFn(torch.autograd.Function):
    """x[rowmask] = 0 (the all-masked guard of the support encoder); gradient is masked the same way."""

    @staticmethod
    def forward(ctx, x, rowmask_u8):
        x = _c(x).clone()
        ops.zero_rows(x, rowmask_u8)
        ctx.save_for_backward(rowmask_u8)
        return x

    @staticmethod
    def backward(ctx, g):
        (rm,) = ctx.saved_tensors
        g = _c(g).clone()
        ops.zero_rows(g, rm)
        return g, None


def zero_rows(x, rowmask_u8):
    return ZeroRowsFn.apply(x, rowmask_u8)


# ------------------------------------------------------------------------------------------------
# criterion
# ------------------------------------------------------------------------------------------------
class LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, coords, labels, vis_u8, target, class_w, w_ce, w_l1):
        losses, total, dl, dc = ops.loss_fwd_bwd(_c(logits), _c(coords), _c(labels).view(-1), _c(vis_u8).view(-1),
                                                 _c(target).view(-1, 2), class_w, w_ce, w_l1, 1.0)
        ctx.save_for_backward(dl, dc)
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)
        return total.view(()), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        if g_total is None:
            return (None,) * 8
        dl, dc = ctx.saved_tensors
        # scaling by the incoming scalar gradient is glue (1/accumulation_steps)
        return dl * g_total, dc * g_total, None, None, None, None, None, None


def cape_loss(logits, coords, labels, vis_u8, target, class_w, w_ce, w_l1):
    return LossFn.apply(logits, coords, labels, vis_u8, target, class_w, float(w_ce), float(w_l1))
