"""Geometry-only support pose-graph encoder (reference `models/geometric_support_encoder.py:22-238`)
on MI355X kernels: coordinate MLP + 2-D/1-D sine encodings -> optional 2x GCN -> 3 post-norm transformer
encoder layers with key-padding mask, including the reference's all-masked guard and the
eval/no-grad nested-tensor behaviour of nn.TransformerEncoder (zeros at padded positions when the mask
is left-aligned over the whole batch)."""
from typing import List

import torch
import torch.nn as nn

from ..hip import functional as HF
from ..hip import ops
from .graph_utils import GCNLayer, adj_from_skeleton
from .positional_encoding import PositionalEncoding1D, SinePositionalEncoding2D


class GeometricSupportEncoder(nn.Module):
    def __init__(self, hidden_dim: int = 256, num_encoder_layers: int = 3, nhead: int = 8, dim_feedforward: int = 1024,
                 dropout: float = 0.1, use_gcn_preenc: bool = False, num_gcn_layers: int = 2, activation: str = "relu"):
        super().__init__()
        if activation != "relu" or hidden_dim != 256 or nhead != 8:
            raise ValueError("MI355X kernels: relu, hidden_dim=256, 8 heads (reference defaults)")
        self.hidden_dim = hidden_dim
        self.use_gcn_preenc = use_gcn_preenc
        self.nhead = nhead
        self.coord_mlp = nn.Sequential(nn.Linear(2, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim))
        self.pos_encoding = SinePositionalEncoding2D(num_feats=hidden_dim // 2, temperature=10000, normalize=True,
                                                     scale=2 * 3.14159265359)
        self.sequence_pos_encoding = PositionalEncoding1D(d_model=hidden_dim, max_len=100, dropout=0.0)
        if use_gcn_preenc:
            self.gcn_layers = nn.ModuleList([GCNLayer(hidden_dim, hidden_dim, kernel_size=2, use_bias=True,
                                                      activation=nn.ReLU(inplace=True), batch_first=True)
                                             for _ in range(num_gcn_layers)])
        else:
            self.gcn_layers = None
        layer = nn.TransformerEncoderLayer(d_model=hidden_dim, nhead=nhead, dim_feedforward=dim_feedforward,
                                           dropout=dropout, activation="relu", batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_encoder_layers)
        self.dropout_p = dropout
        self._streams = [[ops.new_stream_id() for _ in range(4)] for _ in range(num_encoder_layers)]

    def forward(self, support_coords: torch.Tensor, support_mask: torch.Tensor, skeleton_edges: List) -> torch.Tensor:
        bs, num_pts, _ = support_coords.shape
        if num_pts > 100:
            raise RuntimeError("more keypoints than PositionalEncoding1D.max_len=100")
        support_mask = support_mask.bool()
        mask_u8 = ops.as_u8(support_mask)
        m0, m2 = self.coord_mlp[0], self.coord_mlp[2]
        h, pe = HF.support_embed(support_coords, m0.weight, m0.bias, self.sequence_pos_encoding.pe[0].contiguous())
        x = HF.linear(h, m2.weight, m2.bias, residual=pe)                  # coord_emb + pos_emb + seq_pe
        if self.use_gcn_preenc and self.gcn_layers is not None:
            if skeleton_edges is None:
                skeleton_edges = [[] for _ in range(bs)]
            adj = adj_from_skeleton(num_pts, skeleton_edges, mask_u8, support_coords.device)
            for g in self.gcn_layers:
                x = g(x, adj)
        # all-masked guard (geometric_support_encoder.py:201-220): unmask keypoint 0 for the attention, zero the output rows -- one
        # launch (cape_support_masks) instead of nine framework operators.
        # nn.TransformerEncoder's nested-tensor fast path: eval, no grad, mask left-aligned for the whole batch (a host decision:
        # only taken outside autograd, where the reference takes it too)
        fast = False
        if (not self.training) and (not torch.is_grad_enabled()):
            valid = (~support_mask).to(torch.int8)
            valid[:, 0] |= support_mask.all(dim=1).to(torch.int8)          # (keypoint 0 of a fully masked graph counts as valid)
            fast = bool(((valid[:, 1:] - valid[:, :-1]) <= 0).all())
        kpm, zero_u8 = ops.support_masks(mask_u8, pad_rows=fast)
        p = self.dropout_p if self.training else 0.0
        for li, layer in enumerate(self.transformer_encoder.layers):
            sa, st = layer.self_attn, self._streams[li]
            x_a, x_r = HF.fanout(x, 2)              # attention input (q = k = v: one accumulated gradient) | residual
            a = HF.mha(x_a, x_a, x_a, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, self.nhead,
                       mask_mode=2, kpm_u8=kpm, dropout_p=p, rng_stream=st[0])
            x = HF.add_layernorm(x_r, a, layer.norm1.weight, layer.norm1.bias, dropout_p=p, rng_stream=st[1])
            x_f, x_r = HF.fanout(x, 2)
            hdn = HF.ffn(x_f, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, dropout_p=p,
                         rng_stream=st[2])
            x = HF.add_layernorm(x_r, hdn, layer.norm2.weight, layer.norm2.bias, dropout_p=p, rng_stream=st[3])
        # the reference branches on `.any()` (geometric_support_encoder.py:218): that is a device->host sync per step, which
        # stops the host from enqueueing ahead of the GPU; the row-zeroing kernel is a no-op when no row is flagged, so it runs
        # unconditionally instead
        x = HF.zero_rows(x, zero_u8.reshape(-1))
        return x

    def __repr__(self):
        gcn = f", use_gcn_preenc=True ({len(self.gcn_layers)} layers)" if self.use_gcn_preenc else ""
        return f"{self.__class__.__name__}(hidden_dim={self.hidden_dim}, spatial_pe=SinePE2D, sequence_pe=SinePE1D{gcn})"
