"""Deformable-transformer encoder on MI355X kernels.  Module tree and parameter names follow the
reference (`models/deformable_transformer.py:39-291`); the computation is re-laid out for gfx950:

  * tokens stay in one flattened (N, S, 256) fp32 buffer; q = src + pos is emitted by the LayerNorm
    kernel that produces src (no separate add pass);
  * sampling offsets and attention logits come from one (N*S, 384) projection buffer that the MSDA
    kernel consumes directly (softmax over the 16 samples fused into the gather);
  * `self.normK(src + self.dropoutK(x))` is one kernel; `relu + dropout` ride in the GEMM epilogue.
"""
import copy
import math

import torch
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from ..hip import functional as HF
from ..hip import ops


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class MSDeformAttn(nn.Module):
    """Parameters of `MSDeformAttn` (deformable_transformer.py:43-75); forward = :92-114 on HIP kernels."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model != 256 or n_heads != 8:
            raise ValueError("the MI355X MSDA kernel is specialised to d_model=256, 8 heads x 32 channels")
        if n_levels * n_points > 16 or n_levels > 4:
            raise ValueError("n_levels * n_points must be <= 16 and n_levels <= 4")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def project_value(self, input_flatten, padding_rows_u8=None):
        """value_proj(+ zero fill of padded pixels, :95-97): (N, S, 256)."""
        value = HF.linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if padding_rows_u8 is not None:
            value = HF.zero_rows(value, padding_rows_u8)
        return value

    def forward(self, query, reference_points, input_flatten, geo, padding_rows_u8=None, value=None):
        """query (N, Lq, 256); reference_points (N, Lq, L, 2); geo = ops.LevelGeometry.
        `value` may carry a cached value projection (decode)."""
        if reference_points.shape[-1] != 2:
            raise ValueError("Last dim of reference_points must be 2 (box references are not on the CAPE path)")
        assert geo.S == input_flatten.shape[1]
        if value is None:
            value = self.project_value(input_flatten, padding_rows_u8)
        offw = HF.linear_cat2(query, self.sampling_offsets.weight, self.sampling_offsets.bias,
                              self.attention_weights.weight, self.attention_weights.bias)
        out = HF.msda(value, offw, reference_points, geo, self.n_points)
        return HF.linear(out, self.output_proj.weight, self.output_proj.bias)


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if activation != "relu":
            raise ValueError("only relu is fused in the GEMM epilogue (reference default)")
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self._streams = [ops.new_stream_id() for _ in range(3)]

    def forward(self, src, src_pos, pos, reference_points, geo, padding_rows_u8=None):
        """src (N,S,256), src_pos = src + pos.  Returns (new_src, new_src + pos)."""
        p = self.dropout1.p if self.training else 0.0
        # every tensor with two consumers goes through HF.fanout: its gradients are summed by one HIP launch, not by autograd
        src_v, src_r = HF.fanout(src, 2)
        a = self.self_attn(src_pos, reference_points, src_v, geo, padding_rows_u8)
        src = HF.add_layernorm(src_r, a, self.norm1.weight, self.norm1.bias, dropout_p=p, rng_stream=self._streams[0])
        src_f, src_r = HF.fanout(src, 2)
        h = HF.ffn(src_f, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, dropout_p=p,
                   rng_stream=self._streams[1])
        return HF.add_layernorm(src_r, h, self.norm2.weight, self.norm2.bias, pos=pos, dropout_p=p, rng_stream=self._streams[2])


_REF_POINTS = {}        # (level shapes, N, device) -> encoder reference points of an unpadded batch (read-only)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        for layer in self.layers:      # deepcopy duplicated the dropout stream ids
            layer._streams = [ops.new_stream_id() for _ in range(3)]
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(geo, valid_ratios, device):
        """Pixel centres / valid ratio, broadcast to every level (deformable_transformer.py:248-271).
        Constant for a given geometry: host-side setup with torch ops on a few thousand floats."""
        pts = []
        for lvl, (H_, W_) in enumerate(geo.shapes):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            pts.append(torch.stack((ref_x, ref_y), -1))
        ref = torch.cat(pts, 1)
        return (ref[:, :, None] * valid_ratios[:, None]).contiguous()

    def forward(self, src, geo, valid_ratios, pos, padding_rows_u8=None):
        with torch.no_grad():
            if padding_rows_u8 is None:           # unpadded batch: valid ratios are exactly 1, the grid is a constant of the geometry
                key = (tuple(geo.shapes), src.shape[0], str(src.device))
                reference_points = _REF_POINTS.get(key)
                if reference_points is None:
                    reference_points = _REF_POINTS[key] = self.get_reference_points(geo, valid_ratios, src.device)
            else:
                reference_points = self.get_reference_points(geo, valid_ratios, src.device)
        poss = HF.fanout(pos, len(self.layers) + 1)
        output, src0 = HF.fanout(src, 2)
        out_pos = HF.add(src0, poss[0])
        for layer, pos_l in zip(self.layers, poss[1:]):
            output, out_pos = layer(output, out_pos, pos_l, reference_points, geo, padding_rows_u8)
        return output
