"""Deformable transformer (encoder + v1 decoder with support cross-attention) on MI355X kernels.

Mirrors the parameter tree of the reference (`models/deformable_transformer_v2.py:55-370, :951-1131,
:1151-1177`).  Only decoder layer type `v1` exists on the CAPE path (the other variants raise in the
reference) and only it is built here.

MI355X-first differences in *how* (results identical):
  * K/V of the self-attention cache are stored after `in_proj` (the reference caches the pre-projection
    rows and re-projects the whole cache every step, `kv_cache.py:21-36` + nn.MultiheadAttention);
  * the MSDA value projection of the image memory is computed once per layer per episode for decoding
    (the reference recomputes it for every generated token, `deformable_transformer.py:95`);
  * support K/V projections are computed once per episode.
"""
import copy
import math
import os

import numpy as np
import torch
from torch import nn
from torch.nn.init import normal_

from ..hip import functional as HF
from ..hip import ops
from .deformable_transformer import (DeformableTransformerEncoder, DeformableTransformerEncoderLayer, MSDeformAttn)
from ..util.misc import cached_zero_mask
from .kv_cache import KVCache, VCache

_ONES_VR = {}          # (N, levels, device) -> valid ratios of an unpadded batch (read-only)
_FUSED_SELF_ATTN = os.environ.get("CAPE_DEC_SELF_ATTN_NODE", "1") == "1"      # tuning switch: 0 = the round-2 chain of separate nodes


def Embedding(num_embeddings, embedding_dim, padding_idx=None, zero_init=False):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    if zero_init:
        nn.init.constant_(m.weight, 0)
    return m


def get_1d_sincos_pos_embed_from_grid(embed_dim, seq_len):
    pos = np.arange(seq_len, dtype=np.float32)
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


class TransformerDecoderLayer(nn.Module):
    """Decoder layer v1: causal self-attn -> support cross-attn -> MSDA over image memory -> FFN, post-norm."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4,
                 use_qkv_proj=True):
        super().__init__()
        if activation != "relu" or not use_qkv_proj:
            raise ValueError("CAPE path: relu FFN and dec_qkv_proj=True (reference defaults)")
        self.d_model, self.n_heads = d_model, n_heads
        self.attn_q = nn.Linear(d_model, d_model, bias=False)
        self.attn_k = nn.Linear(d_model, d_model, bias=False)
        self.attn_v = nn.Linear(d_model, d_model, bias=False)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.support_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.dropout_support = nn.Dropout(dropout)
        self.norm_support = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.kv_cache = None
        self._streams = [ops.new_stream_id() for _ in range(8)]

    def forward(self, tgt, query_pos, reference_points, src, geo, padding_rows_u8=None, support_features=None,
                support_kpm_u8=None):
        """Teacher-forced pass over the whole sequence (causal mask)."""
        p = self.dropout2.p if self.training else 0.0
        st = self._streams
        # tensors with several consumers go through HF.fanout (one HIP launch sums their gradients)
        sa = self.self_attn
        qp_q, qp_s = HF.fanout(query_pos, 2)
        if _FUSED_SELF_ATTN:
            # attn_q / attn_k / attn_v, MultiheadAttention's in_proj, the causal core and out_proj as one node (:323-341)
            t_a, t_r = HF.fanout(tgt, 2)
            t2 = HF.dec_self_attn(t_a, qp_q, self.attn_q.weight, self.attn_k.weight, self.attn_v.weight, sa.in_proj_weight,
                                  sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, self.n_heads, dropout_p=p, rng_stream=st[0])
        else:
            t_q, t_k, t_v, t_r = HF.fanout(tgt, 4)
            q = HF.linear(t_q, self.attn_q.weight, residual=qp_q)
            k = HF.linear(t_k, self.attn_k.weight)
            v = HF.linear(t_v, self.attn_v.weight)
            t2 = HF.mha(q, k, v, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, self.n_heads,
                        mask_mode=1, dropout_p=p, rng_stream=st[0])
        tgt = HF.add_layernorm(t_r, t2, self.norm2.weight, self.norm2.bias, dropout_p=p, rng_stream=st[1])
        if support_features is not None:
            ca = self.support_attn
            t_q, t_r = HF.fanout(tgt, 2)
            t2 = HF.mha(t_q, support_features, support_features, ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight,
                        ca.out_proj.bias, self.n_heads, mask_mode=2 if support_kpm_u8 is not None else 0,
                        kpm_u8=support_kpm_u8, dropout_p=p, rng_stream=st[2])
            tgt, tgt_pos = HF.add_layernorm(t_r, t2, self.norm_support.weight, self.norm_support.bias, pos=qp_s,
                                            dropout_p=p, rng_stream=st[3])
        else:
            tgt, t_r = HF.fanout(tgt, 2)
            tgt_pos = HF.add(t_r, qp_s)
        t2 = self.cross_attn(tgt_pos, reference_points, src, geo, padding_rows_u8)
        tgt = HF.add_layernorm(tgt, t2, self.norm1.weight, self.norm1.bias, dropout_p=p, rng_stream=st[4])
        t_f, t_r = HF.fanout(tgt, 2)
        h = HF.ffn(t_f, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, dropout_p=p, rng_stream=st[5])
        return HF.add_layernorm(t_r, h, self.norm3.weight, self.norm3.bias, dropout_p=p, rng_stream=st[6])

    # ---- cached single-token step (inference only) -------------------------------------------------
    @torch.no_grad()
    def decode_step(self, tgt, query_pos, reference_points, geo, step, cache):
        """tgt (N,1,256); `cache` = dict(k, v (N, max_len, 256) post-in_proj; value (N,S,256);
        sup_k, sup_v (N,P,256) or None; sup_kpm)."""
        C, H = self.d_model, self.n_heads
        N = tgt.shape[0]
        sa = self.self_attn
        W, B = sa.in_proj_weight, sa.in_proj_bias
        dev = tgt.device
        t2d = tgt.view(N, C)
        tmp = torch.empty(3, N, C, dtype=torch.float32, device=dev)
        ops.gemm(t2d, self.attn_q.weight, tmp[0], N, C, C, residual=query_pos.view(N, C))
        ops.gemm(t2d, self.attn_k.weight, tmp[1], N, C, C)
        ops.gemm(t2d, self.attn_v.weight, tmp[2], N, C, C)
        q = torch.empty(N, 1, C, dtype=torch.float32, device=dev)
        ops.gemm(tmp[0], W, q, N, C, C, bias=B)
        max_len = cache["k"].shape[1]
        # project straight into row `step` of the caches (row stride max_len*C)
        ops.gemm(tmp[1], W[C:], cache["k"][:, step], N, C, C, bias=B[C:], ldc=max_len * C)
        ops.gemm(tmp[2], W[2 * C:], cache["v"][:, step], N, C, C, bias=B[2 * C:], ldc=max_len * C)
        Lk = step + 1
        O, _ = ops.attn_fwd(q, cache["k"], cache["v"], N, H, 1, Lk, (C // H) ** -0.5, mask_mode=0)
        t2 = torch.empty(N, 1, C, dtype=torch.float32, device=dev)
        ops.gemm(O.view(N, C), sa.out_proj.weight, t2, N, C, C, bias=sa.out_proj.bias)
        tgt, _, _, _ = ops.add_layernorm_fwd(tgt, t2, self.norm2.weight, self.norm2.bias)
        tgt_pos = None
        if cache.get("sup_k") is not None:
            ca = self.support_attn
            ops.gemm(tgt.view(N, C), ca.in_proj_weight, q, N, C, C, bias=ca.in_proj_bias)
            P = cache["sup_k"].shape[1]
            O, _ = ops.attn_fwd(q, cache["sup_k"], cache["sup_v"], N, H, 1, P, (C // H) ** -0.5,
                                mask_mode=2 if cache["sup_kpm"] is not None else 0, kpm=cache["sup_kpm"])
            ops.gemm(O.view(N, C), ca.out_proj.weight, t2, N, C, C, bias=ca.out_proj.bias)
            tgt, _, _, tgt_pos = ops.add_layernorm_fwd(tgt, t2, self.norm_support.weight, self.norm_support.bias, pos=query_pos)
        else:
            tgt_pos = ops.add(tgt, query_pos)
        m = self.cross_attn
        LP3 = m.n_heads * m.n_levels * m.n_points
        offw = torch.empty(N, 1, 3 * LP3, dtype=torch.float32, device=dev)
        ops.gemm(tgt_pos.view(N, C), m.sampling_offsets.weight, offw, N, 2 * LP3, C, bias=m.sampling_offsets.bias, ldc=3 * LP3)
        ops.gemm(tgt_pos.view(N, C), m.attention_weights.weight, offw.view(N, -1)[:, 2 * LP3:], N, LP3, C,
                 bias=m.attention_weights.bias, ldc=3 * LP3)
        a = ops.msda_fwd(cache["value"], offw, reference_points, geo, N, 1, m.n_points)
        ops.gemm(a.view(N, C), m.output_proj.weight, t2, N, C, C, bias=m.output_proj.bias)
        tgt, _, _, _ = ops.add_layernorm_fwd(tgt, t2, self.norm1.weight, self.norm1.bias)
        F_ = self.linear1.weight.shape[0]
        h = torch.empty(N, F_, dtype=torch.float32, device=dev)
        ops.gemm(tgt.view(N, C), self.linear1.weight, h, N, F_, C, bias=self.linear1.bias, relu=True)
        ops.gemm(h, self.linear2.weight, t2, N, C, F_, bias=self.linear2.bias)
        tgt, _, _, _ = ops.add_layernorm_fwd(tgt, t2, self.norm3.weight, self.norm3.bias)
        return tgt


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, poly_refine=True, return_intermediate=False, aux_loss=False,
                 query_pos_type="none", vocab_size=None, pad_idx=None, use_anchor=None):
        super().__init__()
        if not poly_refine or query_pos_type != "sine" or use_anchor:
            raise ValueError("CAPE path: with_poly_refine=True, query_pos_type='sine', use_anchor=False (reference defaults)")
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        for layer in self.layers:
            layer._streams = [ops.new_stream_id() for _ in range(8)]
        self.num_layers = num_layers
        self.poly_refine = poly_refine
        self.return_intermediate = return_intermediate
        self.aux_loss = aux_loss
        self.query_pos_type = query_pos_type
        self.coords_embed = None
        self.class_embed = None
        self.pos_trans = None
        self.pos_trans_norm = None
        self.use_anchor = use_anchor
        self.room_class_embed = None
        self.room_class_trans = None
        self.pad_idx = pad_idx
        self.token_embed = Embedding(vocab_size, self.layers[0].d_model, padding_idx=pad_idx, zero_init=False)

    def _seq_embed(self, seq11, seq12, seq21, seq22, delta_x1, delta_x2, delta_y1, delta_y2):
        return HF.token_embed(self.token_embed.weight, self.pad_idx, seq11, seq21, seq12, seq22,
                              delta_x1, delta_x2, delta_y1, delta_y2)

    def _query_pos(self, reference_points):
        qs = HF.query_sine(reference_points)
        qp = HF.linear(qs, self.pos_trans.weight, self.pos_trans.bias)
        return HF.add_layernorm(qp, None, self.pos_trans_norm.weight, self.pos_trans_norm.bias)

    def _mlp(self, mlp, x):
        n = len(mlp.layers)
        for i, layer in enumerate(mlp.layers):
            x = HF.linear(x, layer.weight, layer.bias, relu=(i < n - 1))
        return x

    def forward(self, reference_points, src, geo, src_valid_ratios, padding_rows_u8=None, seq_kwargs=None,
                support_features=None, support_mask=None):
        """Returns (hs (NL,N,L,256), refs (NL,N,L,2), classes (NL,N,L,3)) -- `return_intermediate` layout."""
        if support_features is None:
            support_features = getattr(self, "support_features", None)
        if support_mask is None:
            support_mask = getattr(self, "support_mask", None)
        kpm = ops.as_u8(support_mask) if (support_features is not None and support_mask is not None) else None
        output = self._seq_embed(seq11=seq_kwargs["seq11"], seq12=seq_kwargs["seq12"], seq21=seq_kwargs["seq21"],
                                 seq22=seq_kwargs["seq22"], delta_x1=seq_kwargs["delta_x1"], delta_x2=seq_kwargs["delta_x2"],
                                 delta_y1=seq_kwargs["delta_y1"], delta_y2=seq_kwargs["delta_y2"])
        hs, refs, clss = [], [], []
        nl = len(self.layers)
        mems = HF.fanout(src, nl)                                       # the image memory feeds every layer's value projection
        sups = HF.fanout(support_features, nl) if support_features is not None else (None,) * nl
        for lid, layer in enumerate(self.layers):
            r_scale, r_sine, r_ref = HF.fanout(reference_points, 3)
            ref_in = HF.ref_scale(r_scale, src_valid_ratios)
            query_pos = self._query_pos(r_sine)
            output = layer(output, query_pos, ref_in, mems[lid], geo, padding_rows_u8, sups[lid], kpm)
            o_mlp, o_cls, o_hs, output = HF.fanout(output, 4)
            delta = self._mlp(self.coords_embed[lid], o_mlp)
            reference_points, r_keep = HF.fanout(HF.refine(delta, r_ref), 2)   # no detach between layers (:1096-1102)
            cls = HF.linear(o_cls, self.class_embed[lid].weight, self.class_embed[lid].bias)
            hs.append(o_hs); refs.append(r_keep); clss.append(cls)
        # (the reference stacks the six hidden states, deformable_transformer_v2.py:1128; its only reader takes hs[-1]
        # (roomformer_v2.py:343): the list serves that without a 39 MB copy and the zero-filled gradient of the unread layers)
        return hs, torch.stack(refs), torch.stack(clss)

    @torch.no_grad()
    def decode_step(self, tok, delta, ref_step, geo, valid_ratios, step, caches):
        """One cached AR step.  tok (4,N) int64 [11,12,21,22], delta (4,N) [x1,x2,y1,y2], ref_step (N,1,2)."""
        N = tok.shape[1]
        output = ops.token_embed_fwd(self.token_embed.weight, [tok[0], tok[2], tok[1], tok[3]],
                                     [delta[0], delta[1], delta[2], delta[3]]).view(N, 1, -1)
        ref = ref_step
        for lid, layer in enumerate(self.layers):
            ref_in = ops.ref_scale_fwd(ref, valid_ratios, 1, geo.L).view(N, 1, geo.L, 2)
            qs = ops.query_sine_fwd(ref)
            qp = torch.empty(N, 256, dtype=torch.float32, device=ref.device)
            ops.gemm(qs, self.pos_trans.weight, qp, N, 256, 256, bias=self.pos_trans.bias)
            qp, _, _, _ = ops.add_layernorm_fwd(qp, None, self.pos_trans_norm.weight, self.pos_trans_norm.bias)
            output = layer.decode_step(output, qp.view(N, 1, 256), ref_in, geo, step, caches[lid])
            x = output.view(N, 256)
            mlp = self.coords_embed[lid].layers
            for i, l in enumerate(mlp):
                y = torch.empty(N, l.weight.shape[0], dtype=torch.float32, device=x.device)
                ops.gemm(x, l.weight, y, N, l.weight.shape[0], l.weight.shape[1], bias=l.bias, relu=(i < len(mlp) - 1))
                x = y
            ref = ops.refine_fwd(x.view(N, 1, 2), ref)
        ce = self.class_embed[-1]
        cls = torch.empty(N, ce.weight.shape[0], dtype=torch.float32, device=ref.device)
        ops.gemm(output.view(N, 256), ce.weight, cls, N, ce.weight.shape[0], 256, bias=ce.bias)
        return output, ref, cls


# ------------------------------------------------------------------------------------------------
# fused cached decode step (csrc/decode_step.hip): ~12 launches per layer, LayerNorms applied on load
# ------------------------------------------------------------------------------------------------
def _fold(w_in, w_a):
    """(w_in @ w_a) on the device in exact fp32: the two chained projections of the decoder's self-attention
    (attn_x then MultiheadAttention.in_proj, deformable_transformer_v2.py:323-331) as one weight for inference."""
    out = torch.empty(w_in.shape[0], w_a.shape[1], dtype=torch.float32, device=w_in.device)
    old = ops.get_gemm_precision()
    ops.set_gemm_precision("f32")
    try:
        ops.gemm(w_in, w_a, out, w_in.shape[0], w_a.shape[1], w_in.shape[1], a_mode=0, b_mode=1, ldb=w_a.stride(0))
    finally:
        ops.set_gemm_precision(old)
    return out


class DecodeWeights:
    """Inference-time weights of the fused decode step, rebuilt when any source parameter changed
    (version counter or storage): per layer the folded q|k|v projection (768 x 256) and the concatenated
    sampling_offsets|attention_weights projection (384 x 256)."""

    def __init__(self, decoder):
        self.decoder, self.key, self.layers = decoder, None, None

    def _sources(self):
        out = []
        for l in self.decoder.layers:
            m = l.cross_attn
            out += [l.attn_q.weight, l.attn_k.weight, l.attn_v.weight, l.self_attn.in_proj_weight, m.sampling_offsets.weight,
                    m.sampling_offsets.bias, m.attention_weights.weight, m.attention_weights.bias]
        return out

    def get(self):
        # (the optimizer kernel updates the flat arenas behind autograd's back -- no `_version` bump: its epoch counter, the one
        # the packed GEMM weights follow, is part of the key)
        key = (ops.PackedWeights.epoch,) + tuple((t.data_ptr(), t._version) for t in self._sources())
        if key != self.key:
            C = self.decoder.layers[0].d_model
            layers = []
            with torch.no_grad():
                for l in self.decoder.layers:
                    W = l.self_attn.in_proj_weight
                    m = l.cross_attn
                    layers.append({
                        "w_qkv": torch.cat([_fold(W[:C], l.attn_q.weight), _fold(W[C:2 * C], l.attn_k.weight),
                                            _fold(W[2 * C:], l.attn_v.weight)], 0).contiguous(),
                        "w_off": torch.cat([m.sampling_offsets.weight, m.attention_weights.weight], 0).contiguous(),
                        "b_off": torch.cat([m.sampling_offsets.bias, m.attention_weights.bias], 0).contiguous()})
            self.layers, self.key = layers, key
        return self.layers


def alloc_decode_workspace(N, n_layers, L, dev):
    """Static per-geometry buffers of the fused step (pre-norm sums p1..p4, projections, per-layer query positions)."""
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    return {"emb": f(N, 256), "q": f(N, 256), "qs": f(N, 256), "p1": f(N, 256), "p2": f(N, 256), "p3": f(N, 256),
            "p4": [f(N, 256) for _ in range(n_layers)], "offw": f(N, 384), "h": f(N, 1024),
            "qpos": [None] + [f(N, 256) for _ in range(n_layers - 1)], "refin": [None] + [f(N, L, 2) for _ in range(n_layers - 1)],
            "ref": [None] + [f(N, 2) for _ in range(n_layers)]}


@torch.no_grad()
def decode_step_fused(decoder, dw, ws, caches, geo, vr, step, qpos0, refin0, ref0, out_logits, out_coords, out_hs):
    """One cached AR step for N images.  ws["emb"] holds the embedding of the step's input tokens; layer 0's query
    position embedding / level-scaled points / reference come from per-call tables (qpos0 (256,) broadcast over rows,
    refin0 (N, L, 2), ref0 (N, 2)), the later layers' from the previous layer's tail kernel.  Writes the step's class
    logits, refined coordinates and last hidden state into slot `step` of the output buffers."""
    layers = decoder.layers
    nl = len(layers)
    N = ws["emb"].shape[0]
    C, H = 256, layers[0].n_heads
    scale = (C // H) ** -0.5
    dim_t = ops.dim_t(ws["emb"].device)
    x_prev, ln_prev = ws["emb"], None                       # input of the layer as (pre-norm rows, LayerNorm to apply on load)
    for l, (layer, w, c) in enumerate(zip(layers, dw, caches)):
        qpos = qpos0.view(1, C).expand(N, C) if l == 0 else ws["qpos"][l]
        refin = refin0 if l == 0 else ws["refin"][l]
        ref = ref0 if l == 0 else ws["ref"][l]
        sa = layer.self_attn
        # q | k | v in one launch (folded projections; `+ query_pos` through in_proj_q), k / v straight into the cache row
        ops.decode_linear(x_prev, w["w_qkv"], [ws["q"], c["k"][:, step], c["v"][:, step]], bias=sa.in_proj_bias, in_ln=ln_prev,
                          X2=qpos, W2=sa.in_proj_weight[:C])
        a, _ = ops.attn_fwd(ws["q"].view(N, 1, C), c["k"], c["v"], N, H, 1, step + 1, scale, mask_mode=0)
        ops.decode_linear(a.view(N, C), sa.out_proj.weight, [ws["p1"]], bias=sa.out_proj.bias, res=x_prev, res_ln=ln_prev)
        ln2 = (layer.norm2.weight, layer.norm2.bias)
        if c.get("sup_k") is not None:
            ca = layer.support_attn
            ops.decode_linear(ws["p1"], ca.in_proj_weight[:C], [ws["qs"]], bias=ca.in_proj_bias[:C], in_ln=ln2)
            P = c["sup_k"].shape[1]
            a2, _ = ops.attn_fwd(ws["qs"].view(N, 1, C), c["sup_k"], c["sup_v"], N, H, 1, P, scale,
                                 mask_mode=2 if c["sup_kpm"] is not None else 0, kpm=c["sup_kpm"])
            ops.decode_linear(a2.view(N, C), ca.out_proj.weight, [ws["p2"]], bias=ca.out_proj.bias, res=ws["p1"], res_ln=ln2)
            t_pre, t_ln = ws["p2"], (layer.norm_support.weight, layer.norm_support.bias)
        else:
            t_pre, t_ln = ws["p1"], ln2
        m = layer.cross_attn
        ops.decode_linear(t_pre, w["w_off"], [ws["offw"]], bias=w["b_off"], in_ln=t_ln, in_add=qpos)
        g = ops.msda_fwd(c["value"], ws["offw"].view(N, 1, -1), refin.view(N, 1, geo.L, 2), geo, N, 1, m.n_points)
        ops.decode_linear(g.view(N, C), m.output_proj.weight, [ws["p3"]], bias=m.output_proj.bias, res=t_pre, res_ln=t_ln)
        ln1 = (layer.norm1.weight, layer.norm1.bias)
        ops.decode_linear(ws["p3"], layer.linear1.weight, [ws["h"]], bias=layer.linear1.bias, in_ln=ln1, relu=True)
        ops.decode_linear(ws["h"], layer.linear2.weight, [ws["p4"][l]], bias=layer.linear2.bias, res=ws["p3"], res_ln=ln1)
        ln3 = (layer.norm3.weight, layer.norm3.bias)
        mlp = tuple((q.weight, q.bias) for q in decoder.coords_embed[l].layers)
        last = l == nl - 1
        ops.decode_tail(ws["p4"][l], ln3, mlp, ref, out_coords[:, step] if last else ws["ref"][l + 1], dim_t, vr=vr,
                        cls_head=(decoder.class_embed[l].weight, decoder.class_embed[l].bias) if last else None,
                        cls_out=out_logits[:, step] if last else None,
                        pos_trans=None if last else (decoder.pos_trans.weight, decoder.pos_trans.bias, decoder.pos_trans_norm.weight,
                                                     decoder.pos_trans_norm.bias),
                        qpos_out=None if last else ws["qpos"][l + 1], refin_out=None if last else ws["refin"][l + 1],
                        hs_out=out_hs[:, step] if last else None)
        x_prev, ln_prev = ws["p4"][l], ln3


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", poly_refine=True, return_intermediate_dec=False, aux_loss=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, query_pos_type="none", vocab_size=None, seq_len=1024,
                 pre_decoder_pos_embed=False, learnable_dec_pe=False, dec_attn_concat_src=False, dec_qkv_proj=True,
                 dec_layer_type="v1", pad_idx=None, use_anchor=False, inject_cls_embed=False):
        super().__init__()
        if dec_layer_type != "v1":
            raise TypeError(f"dec_layer_type={dec_layer_type!r}: only the v1 decoder layer accepts support features "
                            "(the other variants raise TypeError in the reference as well)")
        if pre_decoder_pos_embed or dec_attn_concat_src or inject_cls_embed:
            raise ValueError("pre_decoder_pos_embed / dec_attn_concat_src / inject_cls_embed are off on the CAPE path")
        self.d_model, self.nhead = d_model, nhead
        self.poly_refine, self.use_anchor = poly_refine, use_anchor
        self.num_feature_levels = num_feature_levels
        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels,
                                                      nhead, enc_n_points)
        self.encoder = DeformableTransformerEncoder(enc_layer, num_encoder_layers)
        dec_layer = TransformerDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead,
                                            dec_n_points, use_qkv_proj=dec_qkv_proj)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, poly_refine, return_intermediate_dec, aux_loss,
                                          query_pos_type, vocab_size, pad_idx, use_anchor=use_anchor)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.decoder.pos_trans = nn.Linear(d_model, d_model)
        self.decoder.pos_trans_norm = nn.LayerNorm(d_model)
        self.pos_embed = nn.Parameter(torch.zeros(1, seq_len, d_model), requires_grad=learnable_dec_pe)
        self.pos_embed.data.copy_(torch.from_numpy(get_1d_sincos_pos_embed_from_grid(d_model, seq_len)).float().unsqueeze(0))
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    @staticmethod
    def valid_ratio(mask):
        """deformable_transformer_v2.py:157-164 on a (N,h,w) bool mask."""
        _, H, W = mask.shape
        vh = torch.sum(~mask[:, :, 0], 1).float() / H
        vw = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([vw, vh], -1)

    def encode(self, srcs_nhwc, gn_params, masks, has_padding):
        """srcs_nhwc: per-level conv outputs (N,h,w,256) (pre-GroupNorm); gn_params: (gammas, betas).
        Returns the encoder cache dict."""
        geo = ops.LevelGeometry([(s.shape[1], s.shape[2]) for s in srcs_nhwc])
        src_flatten = HF.level_groupnorm(geo, srcs_nhwc, gn_params[0], gn_params[1])
        if has_padding:
            masks_u8 = [m.to(torch.uint8).contiguous() for m in masks]
            with torch.no_grad():
                valid_ratios = torch.stack([self.valid_ratio(m) for m in masks], 1).contiguous()
                pad_rows = torch.cat([m.flatten(1) for m in masks_u8], 1).contiguous()
        else:       # all-False masks: uint8 zeros and valid ratios of exactly 1 (sum(~mask) / H = H / H), one tensor per geometry
            N = srcs_nhwc[0].shape[0]
            masks_u8 = [cached_zero_mask(N, s.shape[1], s.shape[2], s.device, torch.uint8) for s in srcs_nhwc]
            key = (N, len(masks), str(srcs_nhwc[0].device))
            valid_ratios = _ONES_VR.get(key)
            if valid_ratios is None:
                valid_ratios = _ONES_VR[key] = torch.ones(N, len(masks), 2, dtype=torch.float32, device=srcs_nhwc[0].device)
            pad_rows = None
        pos = HF.level_pos(geo, self.level_embed, masks_u8)
        memory = self.encoder(src_flatten, geo, valid_ratios, pos, pad_rows)
        return {"memory": memory, "geo": geo, "valid_ratios": valid_ratios, "pad_rows": pad_rows, "src_flatten": src_flatten}

    def forward(self, enc, query_embed, seq_kwargs, support_features=None, support_mask=None):
        bs = enc["memory"].shape[0]
        L = seq_kwargs["seq11"].shape[1]
        ref_all = HF.sigmoid(query_embed)                                   # (seq_len, 2)
        reference_points = ref_all[:L].unsqueeze(0).expand(bs, -1, -1).contiguous()
        hs, refs, clss = self.decoder(reference_points, enc["memory"], enc["geo"], enc["valid_ratios"], enc["pad_rows"],
                                      seq_kwargs, support_features, support_mask)
        return hs, reference_points, refs, clss

    def _setup_caches(self, max_batch_size, max_seq_length, max_vision_length, model_dim, nhead, dtype, device):
        for layer in self.decoder.layers:
            layer.kv_cache = KVCache(max_batch_size, max_seq_length, model_dim, dtype).to(device)
            layer.cross_attn.cache = VCache(max_batch_size, max_vision_length, nhead, int(model_dim // nhead), dtype).to(device)


def build_deforamble_transformer(args, pad_idx=None):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", poly_refine=args.with_poly_refine, return_intermediate_dec=True, aux_loss=args.aux_loss,
        num_feature_levels=args.num_feature_levels, dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points,
        query_pos_type=args.query_pos_type, vocab_size=args.vocab_size, seq_len=args.seq_len,
        pre_decoder_pos_embed=args.pre_decoder_pos_embed, learnable_dec_pe=args.learnable_dec_pe,
        dec_attn_concat_src=args.dec_attn_concat_src, dec_qkv_proj=args.dec_qkv_proj, dec_layer_type=args.dec_layer_type,
        pad_idx=pad_idx, use_anchor=args.use_anchor, inject_cls_embed=getattr(args, "inject_cls_embed", False))
