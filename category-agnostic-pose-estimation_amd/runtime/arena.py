"""Flat parameter arenas: every trainable tensor of the CAPE model becomes a view into one fp32 slab per
optimizer group (same for its gradient, exp_avg and exp_avg_sq).  With 288 GB of HBM there is no reason
to keep 477 separately allocated tensors: one slab means
  * AdamW + global-norm clipping are two streaming kernels over contiguous memory (no multi-tensor lists),
  * zero_grad is one memset,
  * the data-parallel gradient all-reduce runs on contiguous bucket ranges with no packing copies.
Parameters keep their identity, shape and strides (channels_last conv weights stay channels_last), so
`state_dict()` / `load_state_dict()` are unchanged."""
import torch

# parameters that never receive a gradient on the CAPE path (pinned by the golden
# `no_grad_names` list): torch.optim.AdamW skips them (grad is None), so they stay out of the arenas.
DEAD_PREFIXES = ("support_cross_attention_layers.", "support_attn_layer_norms.", "base_model.room_class_embed.")


def _align(n, a=64):
    return (n + a - 1) // a * a


class ParamGroupArena:
    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += _align(p.numel())          # 256-byte aligned starts: every view is 16-byte aligned for float4 access
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        for p, o in zip(self.params, self.offsets):
            assert p.dtype == torch.float32 and p.device == self.data.device
            self._rebind(p, o)

    def _view(self, slab, p, o):
        # dense tensor with arbitrary (permuted) strides: same strides over the slab
        return torch.as_strided(slab, p.shape, p.stride(), storage_offset=o)

    def _rebind(self, p, o):
        dense = p.numel() == 0 or _is_dense(p)
        assert dense, "parameters must be dense (contiguous up to a permutation)"
        v = self._view(self.data, p, o)
        with torch.no_grad():
            v.copy_(p.data)
        p.data = v
        p.grad = self._view(self.grad, p, o)

    def zero_grad(self):
        self.grad.zero_()

    def grad_view(self, i):
        return self._view(self.grad, self.params[i], self.offsets[i])


def _is_dense(t):
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1), key=lambda x: x[0])
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


def split_groups(model):
    """The reference's two AdamW groups (train_cape_episodic.py:527-538): names without / with 'backbone'."""
    main, backbone, dead = [], [], []
    seen = set()
    for n, p in model.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        if n.startswith(DEAD_PREFIXES):
            dead.append((n, p))
        elif "backbone" in n:
            backbone.append((n, p))
        else:
            main.append((n, p))
    return _colocate(main), backbone, dead


def _colocate(named):
    """Physical order inside an arena is ours to choose (optimizer state and checkpoints go by name).  MSDeformAttn's
    `sampling_offsets` and `attention_weights` always multiply the same rows: their weights are placed back to back
    (and their biases), so that the pair is ONE (384, 256) operand for a single launch (hip/functional.LinearCat2Fn)."""
    out = list(named)
    for i in range(len(out) - 2):
        a, b, c = out[i][0], out[i + 1][0], out[i + 2][0]
        if a.endswith("sampling_offsets.weight") and b.endswith("sampling_offsets.bias") and c.endswith("attention_weights.weight") \
                and a[:-len("sampling_offsets.weight")] == c[:-len("attention_weights.weight")]:
            out[i + 1], out[i + 2] = out[i + 2], out[i + 1]
    return out
