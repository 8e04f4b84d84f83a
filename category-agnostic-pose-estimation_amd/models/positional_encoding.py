"""Support-graph positional encodings (reference `models/positional_encoding.py:7-92`).

`PositionalEncoding1D` owns the `pe` buffer (state_dict key `sequence_pos_encoding.pe`);
`SinePositionalEncoding2D` is parameter-free.  On MI355X both are applied by one kernel
(`cape_support_embed_fwd`) together with the first coordinate-MLP layer."""
import math

import torch
import torch.nn as nn

from ..hip import ops


class PositionalEncoding1D(nn.Module):
    def __init__(self, d_model, max_len=5000, dropout=0.1):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        if self.dropout.p != 0.0 and self.training:
            raise ValueError("dropout on the 1-D positional encoding is not on the CAPE path (reference uses 0.0)")
        return ops.add(x.contiguous(), self.pe[:, : x.size(1), :].expand_as(x).contiguous())


class SinePositionalEncoding2D(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=True, scale=2 * math.pi):
        super().__init__()
        self.num_feats, self.temperature, self.normalize, self.scale = num_feats, temperature, normalize, scale

    def forward_coordinates(self, coord):
        """coord (bs, P, 2) in [0,1] -> (bs, P, 2*num_feats): cat(pos_y, pos_x)."""
        if self.num_feats != 128 or self.temperature != 10000:
            raise ValueError("the MI355X kernel implements num_feats=128, temperature=10000")
        bs, P, _ = coord.shape
        z2 = torch.zeros(256, 2, device=coord.device)
        z1 = torch.zeros(256, device=coord.device)
        zp = torch.zeros(P, 256, device=coord.device)
        _, pe = ops.support_embed_fwd(coord.contiguous(), z2, z1, zp, bs, P, 256)
        return pe.view(bs, P, 256)

    def __repr__(self):
        return (f"{self.__class__.__name__}(num_feats={self.num_feats}, temperature={self.temperature}, "
                f"normalize={self.normalize}, scale={self.scale})")
