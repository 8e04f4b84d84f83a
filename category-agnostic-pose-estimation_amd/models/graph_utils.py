"""Skeleton graph utilities (reference `models/graph_utils.py:15-186`) on MI355X kernels:
`adj_from_skeleton` builds the dual-channel normalised adjacency on device from a flattened edge list,
`GCNLayer` = 1x1 Conv1d (a GEMM) + adjacency aggregation + ReLU."""
import torch
import torch.nn as nn

from ..hip import functional as HF
from ..hip import ops


class DeviceSkeleton:
    """Edge lists already on the device: `edges` int32 (capacity, 2), `start` int32 (bs + 1) offsets into it.
    Lets a captured step (runtime/graph_step.py) refresh the skeletons by copying into fixed buffers."""

    def __init__(self, edges, start):
        self.edges, self.start = edges, start

    def __len__(self):
        return self.start.numel() - 1

    @staticmethod
    def flatten(skeleton):
        flat, start = [], [0]
        for edges in skeleton:
            for e in edges:
                flat.append((int(e[0]), int(e[1])))
            start.append(len(flat))
        return flat, start

    @classmethod
    def from_lists(cls, skeleton, device, capacity=None):
        flat, start = cls.flatten(skeleton)
        cap = max(capacity or 0, len(flat), 1)
        e = torch.zeros(cap, 2, dtype=torch.int32)
        if flat:
            e[:len(flat)] = torch.tensor(flat, dtype=torch.int32)
        return cls(e.to(device), torch.tensor(start, dtype=torch.int32).to(device))


def adj_from_skeleton(num_pts, skeleton, mask, device="cuda"):
    """skeleton: list (len bs) of [[i, j], ...] 0-indexed, or a DeviceSkeleton; mask (bs, num_pts) bool, True = ignore.
    Returns (bs, 2, num_pts, num_pts): [diag(~mask), row-normalised symmetric adjacency]."""
    if isinstance(skeleton, DeviceSkeleton):
        return ops.adjacency(skeleton.edges.contiguous(), skeleton.start, ops.as_u8(mask), len(skeleton), num_pts)
    bs = len(skeleton)
    flat, start = [], [0]
    for edges in skeleton:
        for e in edges:
            flat.append((int(e[0]), int(e[1])))
        start.append(len(flat))
    dev = mask.device if isinstance(mask, torch.Tensor) else torch.device(device)
    edges_t, start_t = _edge_tables(flat, start, dev)[:2]
    return ops.adjacency(edges_t, start_t, ops.as_u8(mask), bs, num_pts)


_EDGE_TABLES = {}


def _edge_tables(flat, start, dev):
    """Device copies of a batch's flattened edge list and row starts.  `torch.tensor(list, device=cuda)` is a synchronous
    pageable copy -- it stalled the host for 1.7 ms twice per training step, waiting for the stream to drain -- so the lists go
    through pinned staging with a non-blocking copy, and recurring skeleton sets (a category's skeleton is fixed) are served
    from a small cache."""
    key = (tuple(flat), tuple(start), str(dev))
    hit = _EDGE_TABLES.get(key)
    if hit is not None:
        return hit
    e_host = torch.tensor(flat if flat else [(0, 0)], dtype=torch.int32)
    s_host = torch.tensor(start, dtype=torch.int32)
    if dev.type == "cuda":
        e_host, s_host = e_host.pin_memory(), s_host.pin_memory()
    out = (e_host.to(dev, non_blocking=True).contiguous(), s_host.to(dev, non_blocking=True))
    if len(_EDGE_TABLES) >= 256:
        _EDGE_TABLES.pop(next(iter(_EDGE_TABLES)))
    _EDGE_TABLES[key] = out + (e_host, s_host)          # the pinned sources stay alive with the entry
    return _EDGE_TABLES[key]


class GCNLayer(nn.Module):
    def __init__(self, in_features, out_features, kernel_size=2, use_bias=True, activation=nn.ReLU(inplace=True),
                 batch_first=True):
        super().__init__()
        if kernel_size != 2 or not isinstance(activation, nn.ReLU) or in_features != out_features:
            raise ValueError("MI355X GCN kernel: kernel_size=2 (self + neighbours), ReLU, square features")
        self.conv = nn.Conv1d(in_features, out_features * kernel_size, kernel_size=1, bias=use_bias)
        self.kernel_size, self.activation, self.batch_first = kernel_size, activation, batch_first

    def forward(self, x, adj):
        assert adj.size(1) == self.kernel_size, \
            f"Adjacency channel dim {adj.size(1)} must match kernel_size {self.kernel_size}"
        if not self.batch_first:
            x = x.transpose(0, 1).contiguous()
        y = HF.linear(x, self.conv.weight.squeeze(-1), self.conv.bias)      # (bs, P, 2*C): channel = k*C + c
        out = HF.gcn_aggregate(y, adj.contiguous())
        return out if self.batch_first else out.transpose(0, 1).contiguous()
