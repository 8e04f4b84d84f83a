"""Raw (non-autograd) launchers: torch tensors in, HIP kernels of libcape_hip.so out.

torch is used for device memory and the current stream only.  Every function checks device,
dtype and layout on the host before handing raw pointers to the kernels (a kernel fault can take the
whole GPU node down), and raises if the library reports an error.  No CPU path exists.
"""
import ctypes
import math
import os
import weakref

import torch

from . import lib

_F32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


_stream_override = [None]        # set by hip/functional._Side: launches go to the weight-gradient side stream
_lazy_fork = [False]             # the side stream has not yet been ordered after the main stream for the current _Side block
_on_fork = [None]                # callback of hip/functional.Runtime: the side stream now holds work that a join must wait for
_wgrad_sink = [None]             # callable(desc, keep, shape) while weight-gradient products are being queued (Runtime.defer_wgrad)


def raw_current_stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    """hipStream_t of torch's current stream on the current device (or the side stream while hip/functional._Side is active).  The two private C entry points cost ~0.5 us; the public
    torch.cuda.current_stream().cuda_stream walks Python helpers for ~8 us -- 40 % of a launcher's host time, ~6 ms per
    training step (1250 launches)."""
    if _stream_override[0] is not None:
        if _lazy_fork[0]:           # first launch of a _Side block that queues its weight gradients: order the side stream now
            _lazy_fork[0] = False
            lib.call("cape_stream_fork", ctypes.c_void_p(raw_current_stream()), ctypes.c_void_p(_stream_override[0]))
            if _on_fork[0] is not None:
                _on_fork[0]()
        return ctypes.c_void_p(_stream_override[0])
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _avail(t):
    """Elements addressable from the tensor's first element to the end of its storage."""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


def _chk(t, name, dtype=_F32, contiguous=True):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on the GPU (no CPU fallback in cape_amd)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")


# ------------------------------------------------------------------------------------------------
# RNG state (device uint64[2] = seed, step) kept as an int64 tensor
# ------------------------------------------------------------------------------------------------
class RngState:
    def __init__(self, seed, device):
        self.t = torch.tensor([seed, 0], dtype=torch.int64, device=device)

    def advance(self):
        lib.call("cape_rng_advance", _p(self.t), _stream())

    @property
    def ptr(self):
        return _p(self.t)


_stream_counter = [0]


def new_stream_id():
    """Distinct dropout stream id per call site instance."""
    _stream_counter[0] += 1
    return _stream_counter[0]


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
_NO_SPLIT_K = os.environ.get("CAPE_NO_SPLIT_K") is not None       # diagnostics: deterministic k order everywhere


def pick_split_k(M, N, K):
    """Enough tiles to fill 256 CUs about twice; each split keeps >= 8 k-tiles of 32."""
    if _NO_SPLIT_K:
        return 1
    if M * N >= 256 * 1024 and K >= 32768:
        # big outputs over a very deep contraction (the FFN weight gradients 1024 x 256 x 43520): the kernel takes 128 x 128
        # tiles here, and two k-splits per CU beat one (profiles/r02_wgrad_sweep.txt: 101 us at split 32 vs 116-123 at 16)
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        return max(8, min(512 // t128, K // 1024) // 8 * 8)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ktiles = (K + 31) // 32
    want = max(1, 1024 // max(tiles, 1))
    sk = int(max(1, min(want, ktiles // 8 if ktiles >= 16 else 1, 64)))
    if sk >= 8:
        sk = sk // 8 * 8          # multiples of 8: the kernel pins each k-split to one XCD (slab read once per L2)
    return sk


class GemmProfiler:
    """Optional per-launch timing of the GEMM family (bench.py's roofline leg): HIP events recorded on the
    stream the kernel is launched on, algorithmic flops = 2*M*N*K per launch."""
    enabled = False
    records = []

    @classmethod
    def start(cls):
        cls.records, cls.enabled = [], True

    @classmethod
    def stop(cls):
        cls.enabled = False
        torch.cuda.synchronize()
        flops = sum(r[2] for r in cls.records)
        ms = sum(r[0].elapsed_time(r[1]) for r in cls.records)
        table = {}
        for r in cls.records:
            t = table.setdefault(r[3], [0, 0.0, 0.0])
            t[0] += 1; t[1] += r[0].elapsed_time(r[1]); t[2] += r[2]
        return {"launches": len(cls.records), "flops": flops, "ms": ms, "table": table, "bytes": sum(r[4] for r in cls.records),
                "products": sum(r[5] for r in cls.records)}

    @staticmethod
    def alg_bytes(M, N, K, nb=1):
        """Algorithmic bytes of one product: A + B + C once each, fp32 (dense formula; an upper bound for the im2col modes)."""
        return 4.0 * nb * (M * K + N * K + M * N)


# GEMM arithmetic: "f32" = exact fp32 MFMA, "bf16x3" = split-bf16 (3 bf16 MFMAs per product, fp32 accumulate).
# Default bf16x3: measured against the reference's golden vectors (profiles/r01_parity_report.json) it deviates by
# 4.9e-5 on logits (tolerance 1e-3), 2.9e-6 on coordinates, identical argmax tokens, 0.3 % on gradient norms, while the
# contraction runs 1.8-2x faster than exact fp32 MFMA.  CAPE_GEMM_PRECISION=f32 selects the exact path (5.8e-6).
_PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}     # "bf16": ONE bf16 MFMA per product -- reported-only (misses the 1e-3 logit bar)
GEMM_PRECISION = _PRECISIONS[os.environ.get("CAPE_GEMM_PRECISION", "bf16x3")]


def set_gemm_precision(name):
    global GEMM_PRECISION
    GEMM_PRECISION = _PRECISIONS[name]


def get_gemm_precision():
    return {v: k for k, v in _PRECISIONS.items()}[GEMM_PRECISION]


class PackedWeights:
    """Registry of fragment-packed weights for the register-stationary GEMM (csrc/gemm_rs.hip, cape_pack_weights).

    A weight that multiplies token rows as the B operand of a dense product with K in {64, 128, 256} is kept, next to its fp32
    master copy, as bf16 (hi, lo) planes in the lane order of the MFMA B fragment: the kernel's per-block weight prologue is
    then K/8 coalesced loads per wave.  Entries are keyed by (storage address, N, K, ldb, b_mode) -- the forward use and the
    dgrad use of one nn.Linear are two entries -- and are valid for one (epoch, tensor version):
      * `invalidate_and_repack()` (called by ArenaAdamW.step(), whose kernel updates the flat arena behind autograd's back)
        re-packs every registered entry with ONE launch over a device-resident item table;
      * any other in-place change (load_state_dict, manual edits) bumps the parameter's `_version`: that entry is re-packed
        alone at its next use.
    Only tensors that are (views of) nn.Parameters are packed: the B operand of every dense a_mode-0 product of the model."""
    enabled = os.environ.get("CAPE_GEMM_PACKED", "1") == "1"
    epoch = 0
    entries = {}
    _table = None            # (device tensor of cape_pack_item, n) rebuilt when entries are added

    @classmethod
    def lookup(cls, B, N, K, ldb, b_mode):
        base = B if isinstance(B, torch.nn.Parameter) else B._base
        if not isinstance(base, torch.nn.Parameter):
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        key = (B.data_ptr(), N, K, int(ldb), b_mode)
        e = cls.entries.get(key)
        if e is not None and e["base"]() is not base:          # the parameter that owned this address is gone: start over
            e = None
        if e is None:
            if capturing:
                return None                                    # never allocate / register inside a graph capture
            nbytes = lib.raw().cape_packed_weight_bytes(N, K)
            e = {"buf": torch.empty(nbytes // 2, dtype=torch.int16, device=B.device), "epoch": -1, "version": -1,
                 "base": weakref.ref(base), "item": (B.data_ptr(), N, K, int(ldb), b_mode)}
            cls.entries[key] = e
            cls._table = None
        if e["epoch"] != cls.epoch or e["version"] != base._version:
            if capturing:
                return None                                    # stale inside a capture: the kernel splits the fp32 weight itself
            cls._pack([e])
            e["epoch"], e["version"] = cls.epoch, base._version
        return e["buf"]

    @classmethod
    def _items(cls, es):
        arr = (lib.PackItem * len(es))()
        for i, e in enumerate(es):
            ptr, N, K, ldb, mode = e["item"]
            arr[i].B, arr[i].out, arr[i].ldb, arr[i].N, arr[i].K, arr[i].b_mode = ptr, e["buf"].data_ptr(), ldb, N, K, mode
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host

    @classmethod
    def _pack(cls, es):
        dev = es[0]["buf"].device
        tab = cls._items(es).to(dev)
        lib.call("cape_pack_weights", _p(tab), len(es), 16, _stream())
        es[0].setdefault("_keep", None)
        es[0]["_keep"] = tab                                    # the table must outlive the asynchronous launch

    @classmethod
    def invalidate_and_repack(cls):
        """All registered weights changed (optimizer step): one launch re-packs them; entries become valid for the new epoch."""
        cls.epoch += 1
        if not cls.enabled or not cls.entries:
            return
        dead = [k for k, e in cls.entries.items() if e["base"]() is None]
        if dead and not torch.cuda.is_current_stream_capturing():
            for k in dead:
                del cls.entries[k]
            cls._table = None
        es = list(cls.entries.values())
        if not es:
            return
        if cls._table is None or cls._table[1] != len(es):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PackedWeights: new weights were registered inside a graph capture")
            cls._table = (cls._items(es).to(es[0]["buf"].device), len(es))
        lib.call("cape_pack_weights", _p(cls._table[0]), len(es), 16, _stream())
        for e in es:
            b = e["base"]()
            e["epoch"], e["version"] = cls.epoch, (b._version if b is not None else -1)

    @classmethod
    def signature(cls):
        """Identity of the registry a captured re-pack launch covers: the entry keys in table order."""
        return tuple(cls.entries.keys())

    @classmethod
    def mark_repacked(cls):
        """A replayed graph ran the optimizer and the re-pack launch of its capture: same bookkeeping as invalidate_and_repack."""
        cls.epoch += 1
        for e in cls.entries.values():
            b = e["base"]()
            e["epoch"], e["version"] = cls.epoch, (b._version if b is not None else -1)

    @classmethod
    def clear(cls):
        cls.entries, cls._table = {}, None


def gemm(A, B, C, M, N, K, a_mode=0, b_mode=0, lda=None, ldb=None, ldc=None, bias=None, scale=None, residual=None,
         ldr=None, relu=False, accumulate=False, split_k=1, dropout_p=0.0, rng=None, rng_stream=0, conv=None,
         colsum_out=None, packed=None, mask_src=None, mask_scale=1.0, batch=None, res_cols=0, bias_strides=None, conv_sub=None):
    """packed: the B operand as fragment-ordered bf16 planes (PackedWeights / cape_pack_weights); looked up automatically when B
    is a parameter (or a view of one) and the product is one the register-stationary kernel takes."""
    for t, n in ((A, "A"), (B, "B"), (C, "C"), (bias, "bias"), (scale, "scale"), (residual, "residual")):
        _chk(t, "gemm." + n, contiguous=False)
    if (_AUTO_SPLIT_NN and a_mode == 0 and b_mode == 1 and split_k == 1 and K >= 512 and M > 64 and batch is None and conv is None
            and bias is None and scale is None and residual is None and not relu and mask_src is None and dropout_p == 0.0
            and colsum_out is None and (ldc is None or ldc == N) and C.is_contiguous() and ((M + 63) // 64) * ((N + 63) // 64) <= 64):
        # plain data gradients with few output tiles over a deep contraction (the support path: 544 rows against K = 512 / 1024 =
        # 36 tiles of 16-32 sequential k-tiles, 21-35 us of pure latency): k-split with atomic accumulation
        sk = pick_split_k(M, N, K)
        if sk >= 4 and _avail(C) >= M * N:
            if not accumulate:
                C.view(-1)[:M * N].zero_()
            split_k, accumulate = sk, True
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.a_mode, d.b_mode = M, N, K, a_mode, b_mode
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.lda = lda if lda is not None else (K if a_mode == 0 else M)
    d.ldb = ldb if ldb is not None else (K if b_mode == 0 else N)
    d.ldc = ldc if ldc is not None else N
    if conv is not None:
        (d.cN, d.cH, d.cW, d.cC, d.cKH, d.cKW, d.cStride, d.cPad, d.cOH, d.cOW, d.cO) = conv
        if conv_sub is not None:          # conv-dgrad over a sub-lattice of the physical filter's taps (cape_gemm_desc, ABI 11)
            (d.cPadX, d.cKHp, d.cKWp, d.cTapH0, d.cTapHS, d.cTapW0, d.cTapWS) = conv_sub
            assert a_mode == 3 and b_mode == 2 and _avail(B) >= d.cO * d.cKHp * d.cKWp * d.cC
            assert _avail(A) >= d.cN * d.cOH * d.cOW * d.cO and _avail(C) >= d.cN * d.cH * d.cW * d.cC
    d.scale = scale.data_ptr() if scale is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldr = ldr if ldr is not None else N
    d.res_cols = int(res_cols)           # residual only on columns < res_cols (0 = all)
    d.relu, d.accumulate, d.split_k = int(relu), int(accumulate), int(split_k)
    d.dropout_p = float(dropout_p)
    d.rng_state = rng.t.data_ptr() if (rng is not None and dropout_p > 0) else None
    d.rng_stream = rng_stream
    d.precision = GEMM_PRECISION
    if (packed is None and PackedWeights.enabled and GEMM_PRECISION >= 1 and a_mode == 0 and b_mode in (0, 1) and batch is None
            and split_k == 1 and K in (64, 128, 256) and M > 64 and 32 <= N <= _PACK_MAX_N):
        packed = PackedWeights.lookup(B, N, K, d.ldb, b_mode)
    if packed is not None and GEMM_PRECISION >= 1:
        assert packed.dtype == torch.int16 and packed.is_cuda and packed.numel() * 2 >= lib.raw().cape_packed_weight_bytes(N, K)
        d.B_packed = packed.data_ptr()
    if mask_src is not None:
        _chk(mask_src, "gemm.mask_src")
        assert mask_src.numel() == M * N and split_k == 1
        d.mask_src, d.ldm, d.mask_scale = mask_src.data_ptr(), N, float(mask_scale)
    if colsum_out is not None:
        _chk(colsum_out, "gemm.colsum_out", contiguous=False)
        assert a_mode == 1 and _avail(colsum_out) >= M       # fused bias gradient of the wgrad product
        d.colsum_out = colsum_out.data_ptr()
    if batch is not None:
        # (count, div, sA0, sA1, sB0, sB1, sC0, sC1): batch b = b0 * div + b1 offsets A/B/C by b0 * s?0 + b1 * s?1 elements
        cnt, div, sA0, sA1, sB0, sB1, sC0, sC1 = batch
        assert cnt >= 1 and div >= 1 and cnt % div == 0
        d.batch, d.batch_div = cnt, div
        d.sA0, d.sA1, d.sB0, d.sB1, d.sC0, d.sC1 = sA0, sA1, sB0, sB1, sC0, sC1
        if bias_strides is not None:
            d.sBias0, d.sBias1 = bias_strides
            assert bias is not None and _avail(bias) >= (cnt // div - 1) * d.sBias0 + (div - 1) * d.sBias1 + N
        last0, last1 = cnt // div - 1, div - 1
        offA, offB, offC = last0 * sA0 + last1 * sA1, last0 * sB0 + last1 * sB1, last0 * sC0 + last1 * sC1
    else:
        offA = offB = offC = 0
    # host-side extent checks (dense modes)
    if a_mode == 0 and M > 0:
        assert _avail(A) >= offA + (M - 1) * d.lda + K, "gemm: A too small"
    if a_mode == 1 and K > 0:
        assert _avail(A) >= offA + (K - 1) * d.lda + M, "gemm: A^T too small"
    if b_mode == 0 and N > 0:
        assert _avail(B) >= offB + (N - 1) * d.ldb + K, "gemm: B too small"
    if b_mode == 1 and K > 0:
        assert _avail(B) >= offB + (K - 1) * d.ldb + N, "gemm: B too small"
    if M > 0:
        assert _avail(C) >= offC + (M - 1) * d.ldc + N, "gemm: C too small"
    if residual is not None:
        assert res_cols % 32 == 0 and 0 <= res_cols <= N
        assert _avail(residual) >= (M - 1) * d.ldr + (res_cols or N), "gemm: residual too small"
    if bias is not None:
        assert _avail(bias) >= N
    if scale is not None:
        assert scale.numel() >= N
    if (_wgrad_sink[0] is not None and a_mode == 1 and accumulate and batch is None and mask_src is None and bias is None
            and b_mode in (1, 3) and _group_ok(d)):
        _wgrad_sink[0](d, (A, B, C, colsum_out), (M, N, K, a_mode, b_mode))      # queued: launched with its group (gemm_group)
        return
    if GemmProfiler.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = _stream()
        e0.record(torch.cuda.ExternalStream(st.value) if _stream_override[0] is not None else None)
        lib.call("cape_gemm_f32", ctypes.byref(d), st)
        e1.record(torch.cuda.ExternalStream(st.value) if _stream_override[0] is not None else None)
        nb = batch[0] if batch is not None else 1
        GemmProfiler.records.append((e0, e1, 2.0 * M * N * K * nb, (M, N, K, a_mode, b_mode, int(split_k), nb),
                                     GemmProfiler.alg_bytes(M, N, K, nb), 1))
        return
    lib.call("cape_gemm_f32", ctypes.byref(d), _stream())


def _group_ok(d):
    """The conditions of cape_gemm_group_f32 (vector path of the tile body) for one weight-gradient product."""
    return (d.A % 16 == 0 and d.B % 16 == 0 and d.lda % 4 == 0 and d.M % 4 == 0 and d.M >= 4 and d.N % 4 == 0 and d.N >= 4
            and (d.b_mode == 3 or d.ldb % 4 == 0) and max(d.lda, d.ldb, d.ldc) < (1 << 31))


_AUTO_SPLIT_NN = os.environ.get("CAPE_AUTO_SPLIT_NN", "1") == "1" and os.environ.get("CAPE_DETERMINISTIC", "0") != "1"
_PACK_MAX_N = int(os.environ.get("CAPE_PACK_MAX_N", "1000000"))      # tuning switch: widest product that reads packed weight planes
_GROUP_BIG_MN = int(os.environ.get("CAPE_GROUP_BIG_MN", str(64 * 1024)))      # tuning switch


def group_tile(M, N, K):
    """Output tile edge of a queued weight gradient: 128 for outputs of >= 256 x 256 over deep contractions (halved L2 -> L1 operand
    traffic per flop, profiles/r02_wgrad_sweep.txt; as a launch of its own a 256 x 256 x 43520 product preferred 64-tiles -- 4 big
    tiles x 64 k-splits of atomics -- but inside a group of ~10 products the splits stay at 8-16: step 25.54 -> 25.34 ms), else 64."""
    return 128 if (M >= 128 and N >= 128 and M * N >= _GROUP_BIG_MN and K >= 4096) else 64


def plan_group_splits(shapes, tile):
    """k-splits for the items [(M, N, K)] of one grouped launch: every block gets about the same number of k-tiles, enough
    blocks to cover the chip (~4 resident 64-tiles or ~2 resident 128-tiles per CU), at least 8 k-tiles per block."""
    target = 1024 if tile == 64 else 512
    tiles = [((M + tile - 1) // tile) * ((N + tile - 1) // tile) for M, N, K in shapes]
    kt = [(K + 31) // 32 for M, N, K in shapes]
    work = sum(t * k for t, k in zip(tiles, kt))
    per_block = max(8.0, work / float(target))              # k-tiles per block
    out = []
    for k in kt:
        s = int(max(1, min(round(k / per_block), k // 8 if k >= 16 else 1, 64)))
        if s >= 8:
            s = s // 8 * 8                                   # multiples of 8: a k-split stays on one XCD (slab read once per L2)
        out.append(s)
    return out


def gemm_group(descs, shapes, tile):
    """descs: list of lib.GemmDesc (weight-gradient products, same b_mode and precision) -> one launch (cape_gemm_group_f32)."""
    n = len(descs)
    assert 1 <= n <= lib.GEMM_GROUP_MAX
    arr = (lib.GemmDesc * n)(*descs)
    if GemmProfiler.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = _stream()
        ext = torch.cuda.ExternalStream(st.value) if _stream_override[0] is not None else None
        e0.record(ext)
        lib.call("cape_gemm_group_f32", arr, n, tile, st)
        e1.record(ext)
        fl = sum(2.0 * M * N * K for M, N, K, _, _ in shapes)
        by = sum(GemmProfiler.alg_bytes(M, N, K) for M, N, K, _, _ in shapes)
        GemmProfiler.records.append((e0, e1, fl, ("group", tile, shapes[0][4], n), by, n))
        return
    lib.call("cape_gemm_group_f32", arr, n, tile, _stream())


def colsum(X, M, N, out, ldx=None, accumulate=True, nbatch=1, batch_stride=0):
    _chk(X, "colsum.X", contiguous=False)
    _chk(out, "colsum.out", contiguous=False)
    ldx = N if ldx is None else ldx
    assert M == 0 or _avail(X) >= (nbatch - 1) * batch_stride + (M - 1) * ldx + N
    assert _avail(out) >= N
    lib.call("cape_colsum_f32", _p(X), ldx, nbatch, batch_stride, M, N, _p(out), int(accumulate), _stream())


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
def add_layernorm_fwd(x, y, gamma, beta, pos=None, dropout_p=0.0, rng=None, rng_stream=0):
    C = x.shape[-1]
    rows = x.numel() // C
    for t, n in ((x, "x"), (y, "y"), (gamma, "gamma"), (beta, "beta"), (pos, "pos")):
        _chk(t, "ln." + n)
    if y is not None:
        assert y.shape == x.shape
    if pos is not None:
        assert pos.numel() == x.numel()
    out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=_F32, device=x.device)
    rstd = torch.empty(rows, dtype=_F32, device=x.device)
    out_pos = torch.empty_like(x) if pos is not None else None
    lib.call("cape_add_layernorm_fwd", _p(x), _p(y), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), _p(pos),
             _p(out_pos), rows, C, float(dropout_p), rng.ptr if (rng is not None and dropout_p > 0) else None,
             rng_stream, _stream())
    return out, mean, rstd, out_pos


def add_layernorm_bwd(d_out, d_out_pos, x, y, gamma, mean, rstd, dgamma, dbeta, dropout_p=0.0, rng=None, rng_stream=0):
    C = x.shape[-1]
    rows = x.numel() // C
    for t, n in ((d_out, "d_out"), (d_out_pos, "d_out_pos"), (x, "x"), (y, "y"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk(t, "ln_bwd." + n)
    d_x = torch.empty_like(x)
    d_y = torch.empty_like(x) if (y is not None and dropout_p > 0) else None
    lib.call("cape_add_layernorm_bwd", _p(d_out), _p(d_out_pos), _p(x), _p(y), _p(gamma), _p(mean), _p(rstd), _p(d_x),
             _p(d_y), _p(dgamma), _p(dbeta), rows, C, float(dropout_p),
             rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())
    return d_x, (d_y if d_y is not None else d_x)


def _gn_workspace(N, C, G, device):
    nbytes = lib.raw().cape_groupnorm_workspace_bytes(N, C, G)
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device), nbytes       # fresh per call: stream-ordered reuse


def groupnorm_fwd(x, gamma, beta, out, out_image_stride, N, HW, C, G=32):
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta")):
        _chk(t, "gn." + n)
    _chk(out, "gn.out", contiguous=False)
    assert x.numel() == N * HW * C
    mean = torch.empty(N * G, dtype=_F32, device=x.device)
    rstd = torch.empty(N * G, dtype=_F32, device=x.device)
    ws, nbytes = _gn_workspace(N, C, G, x.device)
    lib.call("cape_groupnorm_fwd", _p(x), _p(gamma), _p(beta), _p(out), out_image_stride, _p(mean), _p(rstd), N, HW, C, G,
             _p(ws), nbytes, _stream())
    return mean, rstd


def groupnorm_bwd(d_out, d_out_image_stride, x, gamma, mean, rstd, dgamma, dbeta, N, HW, C, G=32):
    _chk(d_out, "gn_bwd.d_out", contiguous=False)
    d_x = torch.empty_like(x)
    ws, nbytes = _gn_workspace(N, C, G, x.device)
    lib.call("cape_groupnorm_bwd", _p(d_out), d_out_image_stride, _p(x), _p(gamma), _p(mean), _p(rstd), _p(d_x),
             _p(dgamma), _p(dbeta), N, HW, C, G, _p(ws), nbytes, _stream())
    return d_x


# ------------------------------------------------------------------------------------------------
# MSDA
# ------------------------------------------------------------------------------------------------
class LevelGeometry:
    """Host-side level table: shapes [(H, W)...], start offsets, S."""

    def __init__(self, shapes):
        self.shapes = [(int(h), int(w)) for h, w in shapes]
        self.L = len(self.shapes)
        starts, s = [], 0
        for h, w in self.shapes:
            starts.append(s)
            s += h * w
        self.S = s
        self.starts = starts
        self._shapes_c = (ctypes.c_int * (2 * self.L))(*[v for hw in self.shapes for v in hw])
        self._starts_c = (ctypes.c_int * self.L)(*starts)


def msda_fwd(value, offw, ref, geo, N, Lq, P=4):
    for t, n in ((value, "value"), (offw, "offw"), (ref, "ref")):
        _chk(t, "msda." + n)
    assert value.numel() == N * geo.S * 256 and offw.numel() == N * Lq * 8 * geo.L * P * 3
    assert ref.numel() == N * Lq * geo.L * 2
    out = torch.empty(N, Lq, 256, dtype=_F32, device=value.device)
    lib.call("cape_msda_fwd", _p(value), _p(offw), _p(ref), geo._shapes_c, geo._starts_c, _p(out), N, geo.S, Lq, geo.L, P,
             _stream())
    return out


MSDA_VALUE_ACCUM = os.environ.get("CAPE_MSDA_VALUE_ACCUM", "auto")          # auto | f64 | fx


def msda_bwd(d_out, value, offw, ref, geo, N, Lq, P=4, need_ref_grad=True, form="split"):
    _chk(d_out, "msda_bwd.d_out")
    d_value = torch.empty_like(value)
    d_offw = torch.empty_like(offw)
    d_ref = torch.empty_like(ref) if need_ref_grad else None
    # "split": LDS-slab scatter + gather kernel (falls back to atomics inside the library when the slab does not fit); its d_value
    # accumulators are fixed-point pairs ("fx") except on the exact-fp32 leg, which keeps the fp64 slab ("f64")
    if form == "split":
        form = MSDA_VALUE_ACCUM if MSDA_VALUE_ACCUM != "auto" else ("f64" if GEMM_PRECISION == 0 else "fx")
    if form == "atomic":
        lib.call("cape_msda_bwd_atomic", _p(d_out), _p(value), _p(offw), _p(ref), geo._shapes_c, geo._starts_c, _p(d_value),
                 _p(d_offw), _p(d_ref), N, geo.S, Lq, geo.L, P, _stream())
    else:
        lib.call("cape_msda_bwd_ex", _p(d_out), _p(value), _p(offw), _p(ref), geo._shapes_c, geo._starts_c, _p(d_value),
                 _p(d_offw), _p(d_ref), N, geo.S, Lq, geo.L, P, {"f64": 0, "fx": 1}[form], _stream())
    return d_value, d_offw, d_ref


# ------------------------------------------------------------------------------------------------
# attention core
# ------------------------------------------------------------------------------------------------
def _ld(t):
    assert t.stride(-1) == 1
    return t.stride(-2)


def attn_fwd(Q, K, V, N, H, Lq, Lk, scale, mask_mode=0, causal_offset=0, kpm=None, dropout_p=0.0, rng=None, rng_stream=0):
    """Q (N,Lq,*) K,V (N,Lk,*) as (possibly strided) views whose last dim holds H*32 head channels."""
    for t, n, L_ in ((Q, "Q", Lq), (K, "K", Lk), (V, "V", Lk)):
        _chk(t, "attn." + n, contiguous=False)
        assert t.dim() == 3 and t.shape[0] == N and t.shape[1] >= L_ and t.shape[2] == H * 32
        assert _avail(t) >= (N - 1) * t.stride(0) + (L_ - 1) * t.stride(1) + H * 32
    if kpm is not None:
        _chk(kpm, "attn.kpm", dtype=torch.uint8)
        assert kpm.numel() == N * Lk
    O = torch.empty(N, Lq, H * 32, dtype=_F32, device=Q.device)
    lse = torch.empty(N, H, Lq, dtype=_F32, device=Q.device)
    lib.call("cape_attn_fwd", _p(Q), _p(K), _p(V), _p(O), _p(lse), _ld(Q), _ld(K), _ld(V), H * 32,
             Q.stride(0), K.stride(0), V.stride(0), Lq * H * 32, N, H, Lq, Lk,
             float(scale), mask_mode, causal_offset, _p(kpm), float(dropout_p),
             rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())
    return O, lse


FLASH_MAX_L = 224


def flash_attn_ok(N, H, Lq, Lk):
    """The fused matrix-core attention (csrc/flash_attn.hip) takes rows of up to 224 keys in the bf16x3 precision mode."""
    return (os.environ.get("CAPE_FLASH_ATTN", "1") == "1" and GEMM_PRECISION == 1 and 32 <= Lq <= FLASH_MAX_L and 32 <= Lk <= FLASH_MAX_L
            and N <= 65535 and H <= 65535)


def _flash_views(Q, K, V, N, H, Lq, Lk):
    for t, n, L_ in ((Q, "Q", Lq), (K, "K", Lk), (V, "V", Lk)):
        _chk(t, "flash_attn." + n, contiguous=False)
        assert t.dim() == 3 and t.shape[0] == N and t.shape[1] >= L_ and t.shape[2] == H * 32 and t.stride(2) == 1
        assert t.stride(1) % 4 == 0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
        assert _avail(t) >= (N - 1) * t.stride(0) + (L_ - 1) * t.stride(1) + H * 32


def flash_attn_fwd(Q, K, V, N, H, Lq, Lk, scale, mask_mode=0, causal_offset=0, kpm=None, dropout_p=0.0, rng=None, rng_stream=0):
    """Q (N, Lq, H*32) / K, V (N, Lk, H*32) possibly strided views -> O (N, Lq, H*32) contiguous, lse (N, H, Lq)."""
    _flash_views(Q, K, V, N, H, Lq, Lk)
    if kpm is not None:
        _chk(kpm, "flash_attn.kpm", dtype=torch.uint8)
        assert kpm.numel() == N * Lk
    O = torch.empty(N, Lq, H * 32, dtype=_F32, device=Q.device)
    lse = torch.empty(N, H, Lq, dtype=_F32, device=Q.device)
    lib.call("cape_flash_attn_fwd", _p(Q), _p(K), _p(V), _p(O), _p(lse), _ld(Q), _ld(K), _ld(V), H * 32, Q.stride(0), K.stride(0),
             V.stride(0), Lq * H * 32, N, H, Lq, Lk, float(scale), mask_mode, causal_offset, _p(kpm), float(dropout_p),
             rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())
    return O, lse


def flash_attn_bwd(dO, Q, K, V, O, lse, dQ, dK, dV, N, H, Lq, Lk, scale, mask_mode=0, causal_offset=0, kpm=None, dropout_p=0.0,
                   rng=None, rng_stream=0):
    _flash_views(Q, K, V, N, H, Lq, Lk)
    _flash_views(dQ, dK, dV, N, H, Lq, Lk)
    _chk(dO, "flash_attn_bwd.dO"); _chk(O, "flash_attn_bwd.O"); _chk(lse, "flash_attn_bwd.lse")
    assert dO.shape == O.shape == (N, Lq, H * 32) and lse.numel() == N * H * Lq
    assert _ld(dQ) == _ld(Q) and _ld(dK) == _ld(K) and _ld(dV) == _ld(V)
    assert dQ.stride(0) == Q.stride(0) and dK.stride(0) == K.stride(0) and dV.stride(0) == V.stride(0)
    ws = torch.empty(N * H * Lq, dtype=_F32, device=dO.device)
    lib.call("cape_flash_attn_bwd", _p(dO), _p(Q), _p(K), _p(V), _p(O), _p(lse), _p(dQ), _p(dK), _p(dV), _p(ws), _ld(Q), _ld(K), _ld(V),
             H * 32, Q.stride(0), K.stride(0), V.stride(0), Lq * H * 32, N, H, Lq, Lk, float(scale), mask_mode, causal_offset, _p(kpm),
             float(dropout_p), rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())


def attn_mm_ok(N, H, Lq, Lk):
    """The matrix-core attention form (batched GEMMs + row softmax) pays off for long rows and needs aligned shapes."""
    return (os.environ.get("CAPE_ATTN_MM", "1") == "1" and Lq >= 64 and Lk >= 64 and Lq % 4 == 0 and Lk % 4 == 0 and N * H <= 65535)


def attn_mm_fwd(Q, K, V, N, H, Lq, Lk, scale, mask_mode=0, causal_offset=0, kpm=None, dropout_p=0.0, rng=None, rng_stream=0):
    """Q K^T and P V as one batched GEMM launch each (a product per (image, head)), row softmax in between.
    Returns O (N, Lq, H*32) and the saved probabilities (P, Pd) for attn_mm_bwd (Pd is P when there is no dropout)."""
    dev = Q.device
    S = torch.empty(N, H, Lq, Lk, dtype=_F32, device=dev)
    gemm(Q, K, S, Lq, Lk, 32, lda=_ld(Q), ldb=_ld(K), ldc=Lk,
         batch=(N * H, H, Q.stride(0), 32, K.stride(0), 32, H * Lq * Lk, Lq * Lk))
    P = torch.empty_like(S)
    Pd = torch.empty_like(S) if dropout_p > 0 else None
    lib.call("cape_attn_softmax_fwd", _p(S), _p(P), _p(Pd), N, H, Lq, Lk, float(scale), mask_mode, causal_offset, _p(kpm),
             float(dropout_p), rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())
    Pu = Pd if Pd is not None else P
    O = torch.empty(N, Lq, H * 32, dtype=_F32, device=dev)
    gemm(Pu, V, O, Lq, 32, Lk, a_mode=0, b_mode=1, lda=Lk, ldb=_ld(V), ldc=H * 32,
         batch=(N * H, H, H * Lq * Lk, Lq * Lk, V.stride(0), 32, Lq * H * 32, 32))
    return O, P, Pu


def attn_mm_bwd(dO, Q, K, V, P, Pu, dQ, dK, dV, N, H, Lq, Lk, scale, dropout_p=0.0, rng=None, rng_stream=0):
    """dV = Pd^T dO ; dPd = dO V^T ; dS = softmax-backward (in place) ; dQ = dS K ; dK = dS^T Q  (batched GEMMs)."""
    blk, img = Lq * Lk, H * Lq * Lk
    gemm(Pu, dO, dV, Lk, 32, Lq, a_mode=1, b_mode=1, lda=Lk, ldb=H * 32, ldc=_ld(dV),
         batch=(N * H, H, img, blk, Lq * H * 32, 32, dV.stride(0), 32))
    dS = torch.empty(N, H, Lq, Lk, dtype=_F32, device=dO.device)
    gemm(dO, V, dS, Lq, Lk, 32, lda=H * 32, ldb=_ld(V), ldc=Lk,
         batch=(N * H, H, Lq * H * 32, 32, V.stride(0), 32, img, blk))
    lib.call("cape_attn_softmax_bwd", _p(P), _p(dS), N, H, Lq, Lk, float(scale), float(dropout_p),
             rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())
    gemm(dS, K, dQ, Lq, 32, Lk, a_mode=0, b_mode=1, lda=Lk, ldb=_ld(K), ldc=_ld(dQ),
         batch=(N * H, H, img, blk, K.stride(0), 32, dQ.stride(0), 32))
    gemm(dS, Q, dK, Lk, 32, Lq, a_mode=1, b_mode=1, lda=Lk, ldb=_ld(Q), ldc=_ld(dK),
         batch=(N * H, H, img, blk, Q.stride(0), 32, dK.stride(0), 32))


def attn_bwd(dO, Q, K, V, O, lse, dQ, dK, dV, N, H, Lq, Lk, scale, mask_mode=0, causal_offset=0, kpm=None, dropout_p=0.0,
             rng=None, rng_stream=0):
    _chk(dO, "attn_bwd.dO")
    for t in (dQ, dK, dV):
        _chk(t, "attn_bwd.grad", contiguous=False)
    assert _ld(dQ) == _ld(Q) and _ld(dK) == _ld(K) and _ld(dV) == _ld(V)
    assert dQ.stride(0) == Q.stride(0) and dK.stride(0) == K.stride(0) and dV.stride(0) == V.stride(0)
    assert O.is_contiguous() and dO.shape == O.shape
    lib.call("cape_attn_bwd", _p(dO), _p(Q), _p(K), _p(V), _p(O), _p(lse), _p(dQ), _p(dK), _p(dV), _ld(Q), _ld(K), _ld(V),
             H * 32, Q.stride(0), K.stride(0), V.stride(0), Lq * H * 32, N, H, Lq, Lk, float(scale), mask_mode, causal_offset, _p(kpm), float(dropout_p),
             rng.ptr if (rng is not None and dropout_p > 0) else None, rng_stream, _stream())


# ------------------------------------------------------------------------------------------------
# elementwise and small ops
# ------------------------------------------------------------------------------------------------
def add(a, b):
    _chk(a, "add.a"); _chk(b, "add.b")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    lib.call("cape_add_f32", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def add_n(tensors):
    """Sum of 1..8 equally shaped contiguous tensors in one pass."""
    k = len(tensors)
    assert 1 <= k <= 8
    for t in tensors:
        _chk(t, "add_n.src")
        assert t.shape == tensors[0].shape
    if k == 1:
        return tensors[0]
    out = torch.empty_like(tensors[0])
    arr = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tensors])
    lib.call("cape_add_n_f32", arr, k, _p(out), out.numel(), _stream())
    return out


def augment_batch(items_dev_u8, n, max_pixels, out_size, mean=None, std=None):
    """items_dev_u8: device uint8 tensor holding n cape_augment_item structs (datasets/transforms.DeviceImagePipeline builds it)."""
    _chk(items_dev_u8, "augment.items", dtype=torch.uint8)
    assert items_dev_u8.numel() >= n * ctypes.sizeof(lib.AugItem)
    if mean is not None:
        _chk(mean, "augment.mean"); _chk(std, "augment.std")
        assert mean.numel() == 3 and std.numel() == 3
    lib.call("cape_augment_batch", _p(items_dev_u8), n, int(max_pixels), int(out_size), _p(mean), _p(std), _stream())


def _row_ld(t, rows, cols):
    if t.is_contiguous():
        return cols
    # leading dims must walk one uniform row stride: (d0, d1, ..., cols) with stride[i] == stride[i+1] * shape[i+1]
    st, sh = t.stride(), t.shape
    assert t.stride(-1) == 1 and all(st[i] == st[i + 1] * sh[i + 1] for i in range(len(sh) - 2)), "add_n_rows: leading dims do not collapse"
    return st[-2]


def add_n_rows(tensors, out=None):
    """Sum of 1..8 tensors of one shape whose last dimension is dense and whose leading dimensions collapse to one row index
    with a uniform row stride (contiguous tensors, or column blocks of wider contiguous buffers): one pass, no copies.
    `out` (same shape; may be strided the same way, may be one of the sources) receives the sum; default: a new tensor."""
    k = len(tensors)
    assert 1 <= k <= 8
    shape = tensors[0].shape
    cols = shape[-1]
    rows = tensors[0].numel() // cols
    lds = []
    for t in tensors:
        _chk(t, "add_n_rows.src", contiguous=False)
        assert t.shape == shape and t.stride(-1) == 1 and cols % 4 == 0
        v = t.reshape(rows, cols) if t.is_contiguous() else t
        if not t.is_contiguous():
            # leading dims must walk one uniform row stride: (d0, d1, ..., cols) with stride[i] == stride[i+1] * shape[i+1]
            st, sh = t.stride(), t.shape
            assert all(st[i] == st[i + 1] * sh[i + 1] for i in range(len(sh) - 2)), "add_n_rows: leading dims do not collapse"
            lds.append(st[-2])
        else:
            lds.append(cols)
        assert _avail(t) >= (rows - 1) * lds[-1] + cols
    if out is None:
        out = torch.empty(shape, dtype=_F32, device=tensors[0].device)
    _chk(out, "add_n_rows.out", contiguous=False)
    assert out.shape == shape
    ldo = _row_ld(out, rows, cols)
    assert _avail(out) >= (rows - 1) * ldo + cols
    arr = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tensors])
    ld = (ctypes.c_longlong * k)(*lds)
    lib.call("cape_add_n_rows_f32", arr, ld, k, _p(out), ldo, rows, cols, _stream())
    return out


def gelu(x):
    _chk(x, "gelu.x")
    out = torch.empty_like(x)
    lib.call("cape_gelu_f32", _p(x), _p(out), x.numel(), _stream())
    return out


def interleave2x2(classes, acc, out):
    """out (N, H, W, C) = (acc or 0) + the four (N, H/2, W/2, C) parity classes [(even,even), (even,odd), (odd,even), (odd,odd)]
    (None = zeros) put back on the full grid (cape_interleave2x2_f32)."""
    _chk(out, "interleave2x2.out"); _chk(acc, "interleave2x2.acc")
    N, H, W, C = out.shape
    assert len(classes) == 4 and H % 2 == 0 and W % 2 == 0 and C % 4 == 0 and (acc is None or acc.shape == out.shape)
    for c in classes:
        _chk(c, "interleave2x2.class")
        assert c is None or c.numel() == N * (H // 2) * (W // 2) * C
    arr = (ctypes.c_void_p * 4)(*[c.data_ptr() if c is not None else None for c in classes])
    lib.call("cape_interleave2x2_f32", arr, _p(acc), _p(out), N, H, W, C, _stream())
    return out


def gelu_bwd(x, g):
    _chk(x, "gelu_bwd.x"); _chk(g, "gelu_bwd.g")
    assert x.shape == g.shape
    dx = torch.empty_like(x)
    lib.call("cape_gelu_bwd_f32", _p(x), _p(g), _p(dx), x.numel(), _stream())
    return dx


def scale_residual_bwd(g, y, gamma, dgamma):
    """(dy, ) of out = x + gamma * y; dgamma (C,) is accumulated into."""
    _chk(g, "scale_residual_bwd.g"); _chk(y, "scale_residual_bwd.y"); _chk(gamma, "scale_residual_bwd.gamma")
    _chk(dgamma, "scale_residual_bwd.dgamma", contiguous=False)
    C = y.shape[-1]
    assert g.shape == y.shape and gamma.numel() == C and _avail(dgamma) >= C
    dy = torch.empty_like(y)
    lib.call("cape_scale_residual_bwd_f32", _p(g), _p(y), _p(gamma), _p(dy), _p(dgamma), y.numel() // C, C, _stream())
    return dy


def scale_residual(x, y, gamma=None):
    """x + y * gamma (per channel, last dim)."""
    _chk(x, "scale_residual.x"); _chk(y, "scale_residual.y"); _chk(gamma, "scale_residual.gamma")
    assert x.shape == y.shape and (gamma is None or gamma.numel() == x.shape[-1])
    out = torch.empty_like(x)
    C = x.shape[-1]
    lib.call("cape_scale_residual_f32", _p(x), _p(y), _p(gamma), _p(out), x.numel() // C, C, _stream())
    return out


def nchw_to_nhwc(x, Cp):
    _chk(x, "nchw_to_nhwc.x")
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, Cp, dtype=_F32, device=x.device)
    lib.call("cape_nchw_to_nhwc", _p(x), _p(out), N, C, H, W, Cp, _stream())
    return out


def bn_fold(w, b, rm, rv, eps=1e-5):
    for t in (w, b, rm, rv):
        _chk(t, "bn_fold")
    scale, shift = torch.empty_like(w), torch.empty_like(w)
    lib.call("cape_bn_fold", _p(w), _p(b), _p(rm), _p(rv), float(eps), _p(scale), _p(shift), w.numel(), _stream())
    return scale, shift


def maxpool3x3s2(x):
    _chk(x, "maxpool.x")
    N, H, W, C = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(N, OH, OW, C, dtype=_F32, device=x.device)
    lib.call("cape_maxpool3x3s2_nhwc", _p(x), _p(out), N, H, W, C, _stream())
    return out


def bn_relu_bwd(dy, y, scale, relu, want_res):
    _chk(dy, "bn_relu_bwd.dy"); _chk(y, "bn_relu_bwd.y"); _chk(scale, "bn_relu_bwd.scale")
    C = dy.shape[-1]
    rows = dy.numel() // C
    d_pre = torch.empty_like(dy)
    d_res = torch.empty_like(dy) if want_res else None
    lib.call("cape_bn_relu_bwd", _p(dy), _p(y), _p(scale), _p(d_pre), _p(d_res), rows, C, int(relu), _stream())
    return d_pre, d_res


def affine_act_(x, scale, bias, residual, relu):
    """In place x = [relu](x * scale + bias [+ residual]) over the last dimension."""
    _chk(x, "affine_act.x"); _chk(scale, "affine_act.scale"); _chk(bias, "affine_act.bias"); _chk(residual, "affine_act.residual")
    C = x.shape[-1]
    assert (scale is None or scale.numel() == C) and (bias is None or bias.numel() == C) and (residual is None or residual.shape == x.shape)
    lib.call("cape_affine_act_f32", _p(x), _p(scale), _p(bias), _p(residual), x.numel() // C, C, int(relu), _stream())
    return x


def relu_drop_bwd(dh, h, inv_keep):
    _chk(dh, "relu_drop_bwd.dh"); _chk(h, "relu_drop_bwd.h")
    out = torch.empty_like(dh)
    lib.call("cape_relu_drop_bwd", _p(dh), _p(h), _p(out), dh.numel(), float(inv_keep), _stream())
    return out


_DIMT = {}


def dim_t(device):
    """10000 ** (2*(k//2)/128), the exact torch expression of the reference (position_encoding.py:33-34)."""
    key = str(device)
    if key not in _DIMT:
        d = torch.arange(128, dtype=torch.float32)
        _DIMT[key] = (10000 ** (2 * (d // 2) / 128)).to(device)
    return _DIMT[key]


def pos_sine_level(mask_u8, level_embed_l, out, out_image_stride, N, h, w, C=256):
    _chk(mask_u8, "pos.mask", dtype=torch.uint8)
    _chk(level_embed_l, "pos.level_embed")
    _chk(out, "pos.out", contiguous=False)
    assert mask_u8.numel() == N * h * w
    lib.call("cape_pos_sine_level", _p(mask_u8), _p(level_embed_l), _p(dim_t(out.device)), _p(out), out_image_stride, N, h,
             w, C, _stream())


def level_embed_add(base, level_embed, geo, out):
    """out = base + level_embed[level of each token] over (N, S, C); see cape_level_embed_add."""
    _chk(base, "level_embed_add.base"); _chk(level_embed, "level_embed_add.level_embed"); _chk(out, "level_embed_add.out")
    N, S, C = base.shape
    assert S == geo.S and out.shape == base.shape and level_embed.shape == (geo.L, C)
    lib.call("cape_level_embed_add", _p(base), _p(level_embed), geo._starts_c, _p(out), N, S, geo.L, C, _stream())


def token_embed_fwd(table, seqs, deltas):
    """seqs = (s11, s21, s12, s22) int64 (R,), deltas = (dx1, dx2, dy1, dy2) float (R,)."""
    _chk(table, "tok.table")
    R = seqs[0].numel()
    for s in seqs:
        _chk(s, "tok.seq", dtype=torch.int64)
        assert s.numel() == R
    for d in deltas:
        _chk(d, "tok.delta")
        assert d.numel() == R
    V, C = table.shape
    out = torch.empty(R, C, dtype=_F32, device=table.device)
    lib.call("cape_token_embed_fwd", _p(table), *[_p(s) for s in seqs], *[_p(d) for d in deltas], _p(out), R, C, V, _stream())
    return out


def token_embed_fwd_into(table, tok_4n, delta_4n, out):
    """tok (4, N) int64 rows [11, 12, 21, 22], delta (4, N) rows [x1, x2, y1, y2] (the decode loop's state layout) -> out (N, C)."""
    _chk(table, "tok.table"); _chk(out, "tok.out"); _chk(tok_4n, "tok.tok", dtype=torch.int64); _chk(delta_4n, "tok.delta")
    R = tok_4n.shape[1]
    V, C = table.shape
    assert tok_4n.shape == (4, R) and delta_4n.shape == (4, R) and out.numel() == R * C
    lib.call("cape_token_embed_fwd", _p(table), _p(tok_4n[0]), _p(tok_4n[2]), _p(tok_4n[1]), _p(tok_4n[3]), _p(delta_4n[0]),
             _p(delta_4n[1]), _p(delta_4n[2]), _p(delta_4n[3]), _p(out), R, C, V, _stream())


def token_embed_bwd(d_out, seqs, deltas, d_table, pad_idx):
    _chk(d_out, "tok_bwd.d_out"); _chk(d_table, "tok_bwd.d_table")
    V, C = d_table.shape
    R = seqs[0].numel()
    lib.call("cape_token_embed_bwd", _p(d_out), *[_p(s) for s in seqs], *[_p(d) for d in deltas], _p(d_table), R, C, V,
             int(pad_idx), _stream())


def query_sine_fwd(ref):
    _chk(ref, "qsine.ref")
    R = ref.numel() // 2
    out = torch.empty(R, 256, dtype=_F32, device=ref.device)
    lib.call("cape_query_sine_fwd", _p(ref), _p(dim_t(ref.device)), _p(out), R, _stream())
    return out


def query_sine_bwd(d_out, ref):
    _chk(d_out, "qsine_bwd.d_out")
    d_ref = torch.empty_like(ref)
    lib.call("cape_query_sine_bwd", _p(d_out), _p(ref), _p(dim_t(ref.device)), _p(d_ref), 0, ref.numel() // 2, _stream())
    return d_ref


def refine_fwd(delta, ref):
    _chk(delta, "refine.delta"); _chk(ref, "refine.ref")
    assert delta.shape == ref.shape
    out = torch.empty_like(ref)
    lib.call("cape_refine_fwd", _p(delta), _p(ref), _p(out), ref.numel(), _stream())
    return out


def refine_bwd(d_new, new_ref, ref):
    _chk(d_new, "refine_bwd.d_new")
    d_delta, d_ref = torch.empty_like(ref), torch.empty_like(ref)
    lib.call("cape_refine_bwd", _p(d_new), _p(new_ref), _p(ref), _p(d_delta), _p(d_ref), 0, ref.numel(), _stream())
    return d_delta, d_ref


def sigmoid_fwd(x):
    _chk(x, "sigmoid.x")
    y = torch.empty_like(x)
    lib.call("cape_sigmoid_fwd", _p(x), _p(y), x.numel(), _stream())
    return y


def sigmoid_bwd(dy, y):
    _chk(dy, "sigmoid_bwd.dy")
    dx = torch.empty_like(y)
    lib.call("cape_sigmoid_bwd", _p(dy), _p(y), _p(dx), 0, y.numel(), _stream())
    return dx


def ref_scale_fwd(ref, valid_ratios, rows_per_image, L):
    _chk(ref, "ref_scale.ref"); _chk(valid_ratios, "ref_scale.vr")
    R = ref.numel() // 2
    out = torch.empty(R, L, 2, dtype=_F32, device=ref.device)
    lib.call("cape_ref_scale_fwd", _p(ref), _p(valid_ratios), _p(out), R, rows_per_image, L, _stream())
    return out


def ref_scale_bwd(d_in, valid_ratios, rows_per_image, L):
    _chk(d_in, "ref_scale_bwd.d_in")
    R = d_in.numel() // (2 * L)
    d_ref = torch.empty(R, 2, dtype=_F32, device=d_in.device)
    lib.call("cape_ref_scale_bwd", _p(d_in), _p(valid_ratios), _p(d_ref), 0, R, rows_per_image, L, _stream())
    return d_ref


def support_embed_fwd(coords, W0, b0, pe1d, N, P, C=256):
    for t in (coords, W0, b0, pe1d):
        _chk(t, "support_embed")
    assert coords.numel() == N * P * 2 and W0.numel() == C * 2 and pe1d.numel() >= P * C
    h = torch.empty(N * P, C, dtype=_F32, device=coords.device)
    pe = torch.empty(N * P, C, dtype=_F32, device=coords.device)
    lib.call("cape_support_embed_fwd", _p(coords), _p(W0), _p(b0), _p(pe1d), _p(dim_t(coords.device)), _p(h), _p(pe), N, P,
             C, _stream())
    return h, pe


def support_embed_bwd(d_h, h, coords, dW0, db0, N, P, C=256):
    for t in (d_h, h, coords, dW0, db0):
        _chk(t, "support_embed_bwd")
    lib.call("cape_support_embed_bwd", _p(d_h), _p(h), _p(coords), _p(dW0), _p(db0), N, P, C, _stream())


def adjacency(edges_i32, edge_start_i32, mask_u8, N, P):
    _chk(edges_i32, "adj.edges", dtype=torch.int32)
    _chk(edge_start_i32, "adj.start", dtype=torch.int32)
    _chk(mask_u8, "adj.mask", dtype=torch.uint8)
    assert edge_start_i32.numel() == N + 1 and mask_u8.numel() == N * P
    adj = torch.empty(N, 2, P, P, dtype=_F32, device=mask_u8.device)
    lib.call("cape_adjacency", _p(edges_i32), _p(edge_start_i32), _p(mask_u8), _p(adj), N, P, _stream())
    return adj


def gcn_aggregate_fwd(y, adj, N, P, C=256):
    _chk(y, "gcn.y"); _chk(adj, "gcn.adj")
    assert y.numel() == N * P * 2 * C and adj.numel() == N * 2 * P * P
    out = torch.empty(N, P, C, dtype=_F32, device=y.device)
    lib.call("cape_gcn_aggregate_fwd", _p(y), _p(adj), _p(out), N, P, C, _stream())
    return out


def gcn_aggregate_bwd(d_out, out, adj, N, P, C=256):
    _chk(d_out, "gcn_bwd.d_out")
    d_y = torch.empty(N, P, 2 * C, dtype=_F32, device=out.device)
    lib.call("cape_gcn_aggregate_bwd", _p(d_out), _p(out), _p(adj), _p(d_y), N, P, C, _stream())
    return d_y


def as_u8(mask):
    """A bool mask as uint8 without a conversion kernel (both are one byte per element)."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


def support_masks(mask_u8, pad_rows=False):
    """(kpm, zero) of the support encoder's key-padding glue (cape_support_masks): mask (N, P) uint8, non-zero = ignore."""
    _chk(mask_u8, "support_masks.mask", dtype=torch.uint8)
    N, P = mask_u8.shape
    kpm, zero = torch.empty_like(mask_u8), torch.empty_like(mask_u8)
    lib.call("cape_support_masks", _p(mask_u8), _p(kpm), _p(zero), N, P, int(pad_rows), _stream())
    return kpm, zero


def zero_rows(x, rowmask_u8):
    _chk(x, "zero_rows.x"); _chk(rowmask_u8, "zero_rows.mask", dtype=torch.uint8)
    C = x.shape[-1]
    rows = x.numel() // C
    assert rowmask_u8.numel() == rows
    lib.call("cape_zero_rows", _p(x), _p(rowmask_u8), rows, C, _stream())


def loss_fwd_bwd(logits, coords, labels, vis_u8, target, class_w, w_ce, w_l1, loss_scale):
    """logits (NL,R,3) coords (NL,R,2) -> losses (2*NL), total (1), d_logits, d_coords."""
    _chk(logits, "loss.logits"); _chk(coords, "loss.coords"); _chk(target, "loss.target"); _chk(class_w, "loss.cw")
    _chk(labels, "loss.labels", dtype=torch.int64); _chk(vis_u8, "loss.vis", dtype=torch.uint8)
    NL = logits.shape[0]
    R = logits.numel() // (NL * 3)
    assert coords.numel() == NL * R * 2 and labels.numel() == R and vis_u8.numel() == R and target.numel() == R * 2
    losses = torch.empty(2 * NL, dtype=_F32, device=logits.device)
    total = torch.empty(1, dtype=_F32, device=logits.device)
    d_logits, d_coords = torch.empty_like(logits), torch.empty_like(coords)
    lib.call("cape_loss_fwd_bwd", _p(logits), _p(coords), _p(labels), _p(vis_u8), _p(target), _p(class_w), float(w_ce),
             float(w_l1), float(loss_scale), _p(losses), _p(total), _p(d_logits), _p(d_coords), NL, R, _stream())
    return losses, total, d_logits, d_coords


def sumsq(g, out_parts):
    """out_parts[b] = partial sum of squares of block b (lib.SUMSQ_PARTS floats; summed in a fixed order by adamw_step)."""
    _chk(g, "sumsq.g"); _chk(out_parts, "sumsq.out", contiguous=False)
    assert out_parts.numel() >= lib.SUMSQ_PARTS and out_parts.stride(-1) == 1
    lib.call("cape_sumsq", _p(g), g.numel(), _p(out_parts), _stream())


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, max_norm, sumsq_t, step_t, lr_dev=None):
    """lr_dev: optional 1-element device tensor holding the learning rate (overrides `lr`; replay-safe)."""
    for t in (p, g, m, v):
        _chk(t, "adamw")
    assert p.numel() == g.numel() == m.numel() == v.numel()
    _chk(step_t, "adamw.step", dtype=torch.int64)
    if lr_dev is not None:
        _chk(lr_dev, "adamw.lr_dev", contiguous=False)
        assert lr_dev.numel() == 1
    n_parts = sumsq_t.numel() if sumsq_t is not None else 0
    lib.call("cape_adamw_step", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
             float(wd), float(max_norm), _p(sumsq_t), n_parts, _p(step_t), _p(lr_dev), _stream())


def step_increment(step_t):
    lib.call("cape_step_increment", _p(step_t), _stream())


def decode_next_tokens(cls_logits, reg, unfinished_i32, tok_i64, delta, step_i32, N, num_bins, min_len, eos, sep, pad):
    _chk(cls_logits, "next.cls"); _chk(reg, "next.reg"); _chk(delta, "next.delta")
    _chk(unfinished_i32, "next.unfinished", dtype=torch.int32); _chk(tok_i64, "next.tok", dtype=torch.int64)
    _chk(step_i32, "next.step", dtype=torch.int32)
    assert cls_logits.numel() == N * 3 and reg.numel() == N * 2 and tok_i64.numel() == 4 * N and delta.numel() == 4 * N
    lib.call("cape_decode_next_tokens", _p(cls_logits), _p(reg), _p(unfinished_i32), _p(tok_i64), _p(delta), _p(step_i32), N,
             num_bins, min_len, eos, sep, pad, _stream())


# ------------------------------------------------------------------------------------------------
# fused decode step (csrc/decode_step.hip)
# ------------------------------------------------------------------------------------------------
def _rows(t, name, width=None):
    """2-D row view checks for the decode kernels: fp32, unit inner stride, 16-byte aligned rows."""
    _chk(t, name, contiguous=False)
    assert t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0, name
    if width is not None:
        assert t.shape[1] == width, (name, tuple(t.shape), width)
    return t


def decode_linear(X, W, outs, bias=None, in_ln=None, in_add=None, X2=None, W2=None, res=None, res_ln=None, relu=False):
    """outs: list of 1..3 (N, seg) row views (any row stride) that tile the Nout = W.shape[0] output columns.
    in_ln / res_ln: (gamma, beta) -> X / res hold pre-norm sums and are layer-normalised on load."""
    N, K = X.shape
    Nout = W.shape[0]
    _rows(X, "decode_linear.X"); _rows(W, "decode_linear.W", K)
    assert 1 <= N <= 64 and 1 <= len(outs) <= 3
    seg = outs[0].shape[1]
    assert seg * len(outs) == Nout
    d = lib.DecodeLinearDesc()
    d.N, d.K, d.Nout = N, K, Nout
    d.X, d.ldx, d.W, d.ldw = X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0)
    keep = [X, W]
    if bias is not None:
        _chk(bias, "decode_linear.bias"); assert bias.numel() == Nout
        d.bias = bias.data_ptr()
    if in_ln is not None:
        g, b = in_ln
        _chk(g, "decode_linear.in_gamma"); _chk(b, "decode_linear.in_beta"); assert g.numel() == K and b.numel() == K
        d.in_gamma, d.in_beta = g.data_ptr(), b.data_ptr()
    if in_add is not None:
        _rows(in_add, "decode_linear.in_add", K); assert in_add.shape[0] == N
        d.in_add, d.ld_add = in_add.data_ptr(), in_add.stride(0)
    if X2 is not None:
        _rows(X2, "decode_linear.X2"); _rows(W2, "decode_linear.W2", X2.shape[1]); assert X2.shape[0] == N
        d.X2, d.ldx2, d.K2, d.W2, d.ldw2, d.n2 = X2.data_ptr(), X2.stride(0), X2.shape[1], W2.data_ptr(), W2.stride(0), W2.shape[0]
    if res is not None:
        _rows(res, "decode_linear.res", Nout); assert res.shape[0] == N
        d.R, d.ldr = res.data_ptr(), res.stride(0)
        if res_ln is not None:
            g, b = res_ln
            _chk(g, "decode_linear.res_gamma"); _chk(b, "decode_linear.res_beta"); assert g.numel() == Nout and b.numel() == Nout
            d.res_gamma, d.res_beta = g.data_ptr(), b.data_ptr()
    d.relu, d.nseg, d.seg = int(relu), len(outs), seg
    for i, o in enumerate(outs):
        _chk(o, "decode_linear.out", contiguous=False)
        assert o.dim() == 2 and o.shape == (N, seg) and o.stride(1) == 1
        assert _avail(o) >= (N - 1) * o.stride(0) + seg
        d.out[i], d.ldo[i] = o.data_ptr(), o.stride(0)
    lib.call("cape_decode_linear", ctypes.byref(d), _stream())


def decode_tail(P4, ln3, mlp, ref, ref_out, dim_t_, vr=None, cls_head=None, cls_out=None, pos_trans=None, qpos_out=None,
                refin_out=None, hs_out=None):
    """mlp = ((W1, b1), (W2, b2), (W3, b3)); cls_head = (Wc, bc) on the last layer; pos_trans = (Wp, bp, gamma, beta) when a
    next layer exists.  ref (N, 2) contiguous; ref_out / cls_out / hs_out row views (N, 2) / (N, ncls) / (N, 256)."""
    N = P4.shape[0]
    _rows(P4, "decode_tail.P4", 256)
    d = lib.DecodeTailDesc()
    d.N = N
    d.P4, d.ldp, d.g3, d.b3 = P4.data_ptr(), P4.stride(0), ln3[0].data_ptr(), ln3[1].data_ptr()
    (W1, b1), (W2, b2), (W3, b3) = mlp
    for w in (W1, W2, W3):
        _chk(w, "decode_tail.W")
    assert W1.shape == (256, 256) and W2.shape == (256, 256) and W3.shape == (2, 256)
    d.W1, d.B1, d.W2, d.B2, d.W3, d.B3 = (t.data_ptr() for t in (W1, b1, W2, b2, W3, b3))
    _chk(ref, "decode_tail.ref"); assert ref.numel() == 2 * N
    d.ref = ref.data_ptr()
    assert ref_out.shape == (N, 2) and ref_out.stride(1) == 1
    d.ref_out, d.ld_ref = ref_out.data_ptr(), ref_out.stride(0)
    d.L = vr.shape[1] if vr is not None else 1
    if cls_head is not None:
        Wc, bc = cls_head
        _chk(Wc, "decode_tail.Wc"); assert Wc.shape[1] == 256 and cls_out.shape == (N, Wc.shape[0]) and cls_out.stride(1) == 1
        d.Wc, d.Bc, d.ncls, d.cls_out, d.ld_cls = Wc.data_ptr(), bc.data_ptr(), Wc.shape[0], cls_out.data_ptr(), cls_out.stride(0)
    if pos_trans is not None:
        Wp, bp, gp, bpn = pos_trans
        _chk(Wp, "decode_tail.Wp"); _chk(qpos_out, "decode_tail.qpos_out"); _chk(refin_out, "decode_tail.refin_out"); _chk(vr, "decode_tail.vr")
        assert Wp.shape == (256, 256) and qpos_out.numel() == N * 256 and refin_out.numel() == N * d.L * 2 and vr.numel() == N * d.L * 2
        d.Wp, d.Bp, d.gp, d.bp = Wp.data_ptr(), bp.data_ptr(), gp.data_ptr(), bpn.data_ptr()
        d.dim_t, d.vr, d.qpos_out, d.refin_out = dim_t_.data_ptr(), vr.data_ptr(), qpos_out.data_ptr(), refin_out.data_ptr()
    if hs_out is not None:
        _rows(hs_out, "decode_tail.hs_out", 256)
        d.hs_out, d.ld_hs = hs_out.data_ptr(), hs_out.stride(0)
    lib.call("cape_decode_tail", ctypes.byref(d), _stream())


class DecodeStepPlan:
    """Persistent descriptor of cape_decode_step (csrc/decode_fused.hip): every pointer that does not change from step to
    step is written once; `launch(step, ...)` fills in the step number, the step's table rows and output slots and launches.
    Keeps the tensors it points to alive."""

    def __init__(self, N, T, geo, n_points, ffn_dim, emb, vr, dim_t_, class_head, pos_trans, layers):
        """layers: list of dicts with the field names of cape_decode_layer_desc -> tensors (sup_mask may be None)."""
        d = lib.DecodeStepDesc()
        self.keep = [emb, vr, dim_t_, class_head, pos_trans, layers]
        assert 1 <= len(layers) <= lib.DECODE_MAX_LAYERS
        _rows(emb, "decode_step.emb", 256); _chk(vr, "decode_step.vr"); _chk(dim_t_, "decode_step.dim_t")
        assert emb.shape == (N, 256) and vr.numel() == N * geo.L * 2 and dim_t_.numel() == 128
        d.N, d.n_layers, d.T, d.S, d.L, d.n_points, d.ffn_dim = N, len(layers), T, geo.S, geo.L, n_points, ffn_dim
        for l, (h, w) in enumerate(geo.shapes):
            d.shapes[2 * l], d.shapes[2 * l + 1] = h, w
            d.level_start[l] = geo.starts[l]
        d.emb, d.vr, d.dim_t = emb.data_ptr(), vr.data_ptr(), dim_t_.data_ptr()
        Wc, bc = class_head
        _chk(Wc, "decode_step.class_w"); _chk(bc, "decode_step.class_b")
        assert Wc.shape[1] == 256 and bc.numel() == Wc.shape[0]
        d.class_w, d.class_b, d.ncls = Wc.data_ptr(), bc.data_ptr(), Wc.shape[0]
        Wp, bp, gp, bpn = pos_trans
        for t_, shp in ((Wp, (256, 256)), (bp, (256,)), (gp, (256,)), (bpn, (256,))):
            _chk(t_, "decode_step.pos_trans"); assert tuple(t_.shape) == shp
        d.pos_w, d.pos_b, d.pos_gamma, d.pos_beta = Wp.data_ptr(), bp.data_ptr(), gp.data_ptr(), bpn.data_ptr()
        P = None
        shapes = {"w_qkv": (768, 256), "b_qkv": (768,), "w_qin": (256, 256), "k_cache": (N, T, 256), "v_cache": (N, T, 256),
                  "w_o": (256, 256), "b_o": (256,), "ln2_g": (256,), "ln2_b": (256,), "w_sq": (256, 256), "b_sq": (256,),
                  "w_so": (256, 256), "b_so": (256,), "lns_g": (256,), "lns_b": (256,), "w_off": (384, 256), "b_off": (384,),
                  "value": (N, geo.S, 256), "w_mo": (256, 256), "b_mo": (256,), "ln1_g": (256,), "ln1_b": (256,),
                  "w1": (ffn_dim, 256), "b1": (ffn_dim,), "w2": (256, ffn_dim), "b2": (256,), "ln3_g": (256,), "ln3_b": (256,),
                  "m1w": (256, 256), "m1b": (256,), "m2w": (256, 256), "m2b": (256,), "m3w": (2, 256), "m3b": (2,)}
        for l, lay in enumerate(layers):
            ld = d.layers[l]
            for name, shp in shapes.items():
                t_ = lay[name]
                _chk(t_, f"decode_step.layers[{l}].{name}")
                assert tuple(t_.shape) == shp, (name, tuple(t_.shape), shp)
                setattr(ld, name, t_.data_ptr())
            for name in ("sup_k", "sup_v"):
                t_ = lay[name]
                _chk(t_, f"decode_step.layers[{l}].{name}")
                assert t_.ndim == 3 and t_.shape[0] == N and t_.shape[2] == 256 and (P is None or t_.shape[1] == P)
                P = t_.shape[1]
                setattr(ld, name, t_.data_ptr())
            m = lay.get("sup_mask")
            if m is not None:
                _chk(m, "decode_step.sup_mask", dtype=torch.uint8); assert tuple(m.shape) == (N, P)
                ld.sup_mask = m.data_ptr()
        d.P = P
        self.d, self.N, self.T, self.L = d, N, T, geo.L

    def launch(self, step, qpos0, refin0, ref0, out_logits, out_coords, out_hs):
        """qpos0 (256,), refin0 (N, L, 2), ref0 (N, 2): layer-0 tables of this step; out_* row views (N, ncls) / (N, 2) / (N, 256)."""
        d, N = self.d, self.N
        assert 0 <= step < self.T
        _chk(qpos0, "decode_step.qpos0"); _chk(refin0, "decode_step.refin0"); _chk(ref0, "decode_step.ref0")
        assert qpos0.numel() == 256 and refin0.numel() == N * self.L * 2 and ref0.numel() == N * 2
        for t_, w in ((out_logits, d.ncls), (out_coords, 2), (out_hs, 256)):
            if not t_.is_cuda or t_.dtype != _F32 or t_.shape != (N, w) or t_.stride(1) != 1:
                raise ValueError("decode_step: output slots must be (N, width) fp32 row views on the GPU")
            if _avail(t_) < (N - 1) * t_.stride(0) + w:
                raise ValueError("decode_step: output slot exceeds its storage")
        d.step = step
        d.qpos0, d.refin0, d.ref0 = qpos0.data_ptr(), refin0.data_ptr(), ref0.data_ptr()
        d.out_logits, d.ld_logits = out_logits.data_ptr(), out_logits.stride(0)
        d.out_coords, d.ld_coords = out_coords.data_ptr(), out_coords.stride(0)
        d.out_hs, d.ld_hs = out_hs.data_ptr(), out_hs.stride(0)
        lib.call("cape_decode_step", ctypes.byref(d), _stream())


def decode_advance(cls_slot, reg_slot, unfinished_i32, tok_i64, delta, step, N, num_bins, min_len, eos, sep, pad, table=None,
                   embed_out=None, alive_out=None):
    """cls_slot (N, 3) / reg_slot (N, 2) row views (e.g. out_logits[:, i]); see cape_decode_advance."""
    assert cls_slot.shape == (N, 3) and reg_slot.shape == (N, 2) and cls_slot.stride(1) == 1 and reg_slot.stride(1) == 1
    _chk(unfinished_i32, "advance.unfinished", dtype=torch.int32); _chk(tok_i64, "advance.tok", dtype=torch.int64); _chk(delta, "advance.delta")
    assert tok_i64.numel() == 4 * N and delta.numel() == 4 * N and unfinished_i32.numel() == N
    C = vocab = 0
    if embed_out is not None:
        _chk(table, "advance.table"); _chk(embed_out, "advance.embed_out")
        vocab, C = table.shape
        assert embed_out.numel() == N * C
    if alive_out is not None:
        _chk(alive_out, "advance.alive", dtype=torch.int32, contiguous=False)
    lib.call("cape_decode_advance", _p(cls_slot), cls_slot.stride(0), _p(reg_slot), reg_slot.stride(0), _p(unfinished_i32),
             _p(tok_i64), _p(delta), int(step), N, num_bins, min_len, eos, sep, pad, _p(table), vocab, C, _p(embed_out),
             _p(alive_out), _stream())
