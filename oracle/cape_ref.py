"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CAPE hot path (the parity oracle).

This file restates, in plain functional `torch` CPU ops over a flat `state_dict`
(reference key names), the algorithm of the reference's episodic training /
inference path.  It is the *checker* for the HIP product path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The
product package never imports anything under `oracle/` and fails loudly when its
HIP library is missing -- there is no CPU fallback through this file.

Pinning: `tests/test_oracle_golden.py` checks every function below against golden
vectors produced by the *real* reference (imported in the build container through
`oracle/refshim.py`, script `oracle/make_golden.py`, fixtures `tests/golden/*.npz`).

Each function cites the reference file:line (relative to /root/reference) it follows.
Third-party arithmetic that is not under /root/reference: torchvision ResNet-50
(v1.5 architecture restated from the published layout; parity unpinned upstream, pinned
here against the shim's restatement -- see SURVEY.md section 8c) and torch.nn
(MultiheadAttention / TransformerEncoderLayer / LayerNorm / GroupNorm / grid_sample),
restated from their documented formulas and pinned through the golden vectors.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# configuration (defaults = models/train_cape_episodic.py:86-254)
# ----------------------------------------------------------------------------


class Cfg:
    def __init__(self, **kw):
        self.hidden_dim = 256
        self.nheads = 8
        self.enc_layers = 6
        self.dec_layers = 6
        self.dim_feedforward = 1024
        self.num_feature_levels = 4
        self.n_points = 4
        self.seq_len = 200
        self.num_bins = 44                 # int(sqrt(vocab_size=2000)), mp100_cape.py:118-121
        self.patch_size = 1                # roomformer_v2.py:995
        self.dropout = 0.1
        self.support_encoder_layers = 3
        self.num_gcn_layers = 2
        self.use_gcn_preenc = True
        self.cls_loss_coef = 1.0
        self.coords_loss_coef = 5.0
        self.room_cls_loss_coef = 0.0
        self.eos_weight = 20.0
        self.semantic_classes = 70
        self.__dict__.update(kw)

    # tokenizer ids, datasets/discrete_tokenizer.py:16-28
    @property
    def bos(self): return self.num_bins * self.num_bins
    @property
    def eos(self): return self.num_bins * self.num_bins + 1
    @property
    def sep(self): return self.num_bins * self.num_bins + 2
    @property
    def pad(self): return self.num_bins * self.num_bins + 3


def _drop(x, p, train):
    return F.dropout(x, p, True) if (train and p > 0) else x


def linear(x, sd, name, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


def layer_norm(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


# ----------------------------------------------------------------------------
# backbone: models/backbone.py:13-97 + torchvision resnet50 (v1.5, published layout)
# ----------------------------------------------------------------------------

def frozen_bn(x, sd, name):
    """models/backbone.py:32-40 (eps 1e-5, scale = w * rsqrt(var + eps))."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    rv, rm = sd[name + ".running_var"], sd[name + ".running_mean"]
    scale = w * (rv + 1e-5).rsqrt()
    bias = b - rm * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def _bottleneck(x, sd, p, stride):
    out = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = F.relu(frozen_bn(F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2"))
    out = frozen_bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    if (p + "downsample.0.weight") in sd:
        x = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
    return F.relu(out + x)


def resnet50_body(x, sd, prefix="base_model.backbone.0.body."):
    """Returns [C3, C4, C5] (layer2/3/4 outputs), backbone.py:47-57."""
    x = F.relu(frozen_bn(F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3), sd, prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (nblk, stride) in enumerate([(3, 1), (4, 2), (6, 2), (3, 2)], start=1):
        for b in range(nblk):
            x = _bottleneck(x, sd, f"{prefix}layer{li}.{b}.", stride if b == 0 else 1)
        if li >= 2:
            feats.append(x)
    return feats


def position_embedding_sine(mask, num_pos_feats=128, temperature=10000.0):
    """models/position_encoding.py:22-40 (normalize=True, scale 2*pi, eps 1e-6). mask (N,h,w) bool."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3)          # (N,h,w,256) channels-last


def image_features(images, sd, cfg, img_mask=None):
    """Backbone + input_proj + sine pos + flatten (roomformer_v2.py:299-327,
    deformable_transformer_v2.py:185-205).  Returns src (N,S,256), lvl_pos (N,S,256),
    mask_flat (N,S), spatial_shapes [(h,w)...], valid_ratios (N,L,2)."""
    N, _, H, W = images.shape
    if img_mask is None:
        img_mask = torch.zeros(N, H, W, dtype=torch.bool)
    feats = resnet50_body(images, sd)
    ip = "base_model.input_proj."
    srcs, masks = [], []
    ps = cfg.patch_size
    for l, f in enumerate(feats):
        src = F.conv2d(f, sd[f"{ip}{l}.0.weight"], sd[f"{ip}{l}.0.bias"], stride=ps)
        src = F.group_norm(src, 32, sd[f"{ip}{l}.1.weight"], sd[f"{ip}{l}.1.bias"], 1e-5)
        m = F.interpolate(img_mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
        if ps != 1:
            m = F.interpolate(m[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
        srcs.append(src)
        masks.append(m)
    for l in range(len(feats), cfg.num_feature_levels):
        inp = feats[-1] if l == len(feats) else srcs[-1]
        if ps == 1:
            src = F.conv2d(inp, sd[f"{ip}{l}.0.weight"], sd[f"{ip}{l}.0.bias"], stride=2, padding=1)
        else:
            src = F.conv2d(inp, sd[f"{ip}{l}.0.weight"], sd[f"{ip}{l}.0.bias"], stride=2 * ps)
        src = F.group_norm(src, 32, sd[f"{ip}{l}.1.weight"], sd[f"{ip}{l}.1.bias"], 1e-5)
        m = F.interpolate(img_mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
        srcs.append(src)
        masks.append(m)
    level_embed = sd["base_model.transformer.level_embed"]
    src_flat, pos_flat, mask_flat, shapes, ratios = [], [], [], [], []
    for l, (s, m) in enumerate(zip(srcs, masks)):
        n, c, h, w = s.shape
        shapes.append((h, w))
        src_flat.append(s.flatten(2).transpose(1, 2))
        pos = position_embedding_sine(m).reshape(n, h * w, c)
        pos_flat.append(pos + level_embed[l].view(1, 1, -1))
        mask_flat.append(m.flatten(1))
        vh = (~m[:, :, 0]).sum(1).float() / h          # deformable_transformer_v2.py:157-164
        vw = (~m[:, 0, :]).sum(1).float() / w
        ratios.append(torch.stack([vw, vh], -1))
    return (torch.cat(src_flat, 1), torch.cat(pos_flat, 1), torch.cat(mask_flat, 1), shapes,
            torch.stack(ratios, 1))


# ----------------------------------------------------------------------------
# multi-scale deformable attention: models/deformable_transformer.py:76-141
# ----------------------------------------------------------------------------

def msda_core(value, shapes, loc, attw):
    """value (N,S,M,D); loc (N,Lq,M,L,P,2) normalised (x,y); attw (N,Lq,M,L,P) softmaxed.
    Bilinear, zeros padding, align_corners=False: pixel = loc*size - 0.5 (restated
    without grid_sample so that this function is an independent statement)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(N, Lq, M, D, dtype=value.dtype)
    start = 0
    nidx = torch.arange(N).view(N, 1, 1, 1)
    midx = torch.arange(M).view(1, 1, M, 1)
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]               # (N,HW,M,D)
        start += H * W
        x = loc[:, :, :, l, :, 0] * W - 0.5             # (N,Lq,M,P)
        y = loc[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        fx, fy = x - x0, y - y0
        x0, y0 = x0.long(), y0.long()
        for dy, dx, wgt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)),
                            (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1))
            g = v[nidx, idx, midx]                      # (N,Lq,M,P,D)
            wt = (wgt * ok.to(value.dtype) * attw[:, :, :, l, :])
            out = out + (g * wt[..., None]).sum(3)
    return out.reshape(N, Lq, M * D)


def msda(query, ref, src, shapes, pad_mask, sd, name, cfg):
    """MSDeformAttn.forward, deformable_transformer.py:92-114. ref (N,Lq,L,2)."""
    N, Lq, C = query.shape
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.n_points
    value = linear(src, sd, name + ".value_proj")
    if pad_mask is not None:
        value = value.masked_fill(pad_mask[..., None], 0.0)
    value = value.view(N, -1, M, C // M)
    off = linear(query, sd, name + ".sampling_offsets").view(N, Lq, M, L, P, 2)
    aw = linear(query, sd, name + ".attention_weights").view(N, Lq, M, L * P)
    aw = F.softmax(aw, -1).view(N, Lq, M, L, P)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=query.dtype)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msda_core(value, shapes, loc, aw)
    return linear(out, sd, name + ".output_proj")


def encoder_reference_points(shapes, valid_ratios):
    """deformable_transformer.py:248-271."""
    pts = []
    for l, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, l, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, l, 0] * W)
        pts.append(torch.stack((rx, ry), -1))
    ref = torch.cat(pts, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def encoder(src, pos, mask_flat, shapes, valid_ratios, sd, cfg, train=False):
    """6 x DeformableTransformerEncoderLayer, deformable_transformer.py:212-231, :287-291."""
    ref = encoder_reference_points(shapes, valid_ratios)
    p = cfg.dropout
    x = src
    for i in range(cfg.enc_layers):
        n = f"base_model.transformer.encoder.layers.{i}"
        a = msda(x + pos, ref, x, shapes, mask_flat, sd, n + ".self_attn", cfg)
        x = layer_norm(x + _drop(a, p, train), sd, n + ".norm1")
        h = linear(_drop(F.relu(linear(x, sd, n + ".linear1")), p, train), sd, n + ".linear2")
        x = layer_norm(x + _drop(h, p, train), sd, n + ".norm2")
    return x


# ----------------------------------------------------------------------------
# attention (torch.nn.MultiheadAttention formula) and the decoder
# ----------------------------------------------------------------------------

def mha(q_in, k_in, v_in, sd, name, nheads, attn_mask=None, key_padding_mask=None, p=0.0, train=False):
    """nn.MultiheadAttention, batch-major inputs (N,Lq,C)/(N,Lk,C); own in_proj + out_proj;
    q scaled by head_dim**-0.5; additive float attn_mask (Lq,Lk); boolean key_padding_mask
    (N,Lk) True=ignore; dropout on the attention probabilities."""
    N, Lq, C = q_in.shape
    Lk = k_in.shape[1]
    W, B = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    d = C // nheads
    q = F.linear(q_in, W[:C], B[:C]).view(N, Lq, nheads, d).transpose(1, 2)
    k = F.linear(k_in, W[C:2 * C], B[C:2 * C]).view(N, Lk, nheads, d).transpose(1, 2)
    v = F.linear(v_in, W[2 * C:], B[2 * C:]).view(N, Lk, nheads, d).transpose(1, 2)
    s = (q * (d ** -0.5)) @ k.transpose(-1, -2)        # (N,h,Lq,Lk)
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = _drop(F.softmax(s, -1), p, train)
    o = (a @ v).transpose(1, 2).reshape(N, Lq, C)
    return linear(o, sd, name + ".out_proj")


def seq_embed(sd, t):
    """TransformerDecoder._seq_embed, deformable_transformer_v2.py:984-997."""
    tab = sd["base_model.transformer.decoder.token_embed.weight"]
    e11, e21, e12, e22 = tab[t["seq11"]], tab[t["seq21"]], tab[t["seq12"]], tab[t["seq22"]]
    dx1, dx2 = t["delta_x1"][..., None], t["delta_x2"][..., None]
    dy1, dy2 = t["delta_y1"][..., None], t["delta_y2"][..., None]
    return e11 * dx2 * dy2 + e21 * dx1 * dy2 + e12 * dx2 * dy1 + e22 * dx1 * dy1


def query_pos_sine(ref):
    """get_query_pos_embed, deformable_transformer_v2.py:1005-1018: x-block then y-block."""
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * (dim_t // 2) / 128)
    pos = (ref * (2 * math.pi))[:, :, :, None] / dim_t
    return torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=4).flatten(2)


def inverse_sigmoid(x, eps=1e-5):
    """util/misc.py:436-440."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def mlp3(x, sd, name):
    """roomformer_v2.py:956-968 with 3 layers."""
    x = F.relu(linear(x, sd, name + ".layers.0"))
    x = F.relu(linear(x, sd, name + ".layers.1"))
    return linear(x, sd, name + ".layers.2")


def decoder_layer(tgt, qpos, ref_in, memory, shapes, mask_flat, self_k_src, attn_mask, support, support_mask,
                  sd, n, cfg, train=False):
    """TransformerDecoderLayer v1, deformable_transformer_v2.py:320-370.
    `self_k_src` = the tokens whose attn_k/attn_v projections form keys/values (== tgt when
    teacher-forcing, the cached prefix when decoding)."""
    p = cfg.dropout
    q = F.linear(tgt, sd[n + ".attn_q.weight"]) + qpos
    k = F.linear(self_k_src, sd[n + ".attn_k.weight"])
    v = F.linear(self_k_src, sd[n + ".attn_v.weight"])
    t2 = mha(q, k, v, sd, n + ".self_attn", cfg.nheads, attn_mask=attn_mask, p=p, train=train)
    tgt = layer_norm(tgt + _drop(t2, p, train), sd, n + ".norm2")
    if support is not None:
        t2 = mha(tgt, support, support, sd, n + ".support_attn", cfg.nheads,
                 key_padding_mask=support_mask, p=p, train=train)
        tgt = layer_norm(tgt + _drop(t2, p, train), sd, n + ".norm_support")
    t2 = msda(tgt + qpos, ref_in, memory, shapes, mask_flat, sd, n + ".cross_attn", cfg)
    tgt = layer_norm(tgt + _drop(t2, p, train), sd, n + ".norm1")
    h = linear(_drop(F.relu(linear(tgt, sd, n + ".linear1")), p, train), sd, n + ".linear2")
    return layer_norm(tgt + _drop(h, p, train), sd, n + ".norm3")


def decoder(tgt_embed, ref0, memory, shapes, valid_ratios, mask_flat, support, support_mask, sd, cfg,
            attn_mask, kv_prefix=None, train=False):
    """TransformerDecoder.forward, deformable_transformer_v2.py:1024-1131 (poly_refine, sine query
    pos, aux classes).  ref0 (N,L,2).  kv_prefix: optional list (per layer) of previous layer-input
    tokens (N,i,256) for cached decoding; returns the per-layer inputs so the caller can extend it."""
    out = tgt_embed
    ref = ref0
    hs, refs, clss, layer_inputs = [], [], [], []
    dn = "base_model.transformer.decoder"
    for lid in range(cfg.dec_layers):
        ref_in = ref[:, :, None] * valid_ratios[:, None]
        qpos = layer_norm(linear(query_pos_sine(ref), sd, dn + ".pos_trans"), sd, dn + ".pos_trans_norm")
        layer_inputs.append(out)
        ksrc = out if kv_prefix is None else torch.cat([kv_prefix[lid], out], 1)
        out = decoder_layer(out, qpos, ref_in, memory, shapes, mask_flat, ksrc, attn_mask, support,
                            support_mask, sd, f"{dn}.layers.{lid}", cfg, train)
        ref = torch.sigmoid(mlp3(out, sd, f"base_model.coords_embed.{lid}") + inverse_sigmoid(ref))
        hs.append(out)
        refs.append(ref)
        clss.append(linear(out, sd, f"base_model.class_embed.{lid}"))
    return torch.stack(hs), torch.stack(refs), torch.stack(clss), layer_inputs


def causal_mask(L):
    """roomformer_v2.py:275-283: 0 on/below the diagonal, -inf above."""
    return torch.triu(torch.full((L, L), float("-inf")), diagonal=1)


# ----------------------------------------------------------------------------
# geometric support encoder: models/geometric_support_encoder.py:134-228
# ----------------------------------------------------------------------------

def adj_from_skeleton(P, skeleton, mask):
    """models/graph_utils.py:46-80.  mask (N,P) True=ignore.  Returns (N,2,P,P)."""
    N = len(skeleton)
    adj = torch.zeros(N, P, P)
    for b in range(N):
        for e in skeleton[b]:
            i, j = int(e[0]), int(e[1])
            if i < P and j < P:                       # graph_utils.py:59-60 (negative indices wrap as in torch)
                adj[b, i, j] = 1
    adj = torch.maximum(adj, adj.transpose(1, 2))       # symmetrise (:67-69 equals elementwise max for 0/1)
    keep = (~mask).float()
    adj = adj * keep[:, :, None] * keep[:, None, :]
    adj = torch.nan_to_num(adj / adj.sum(-1, keepdim=True))
    return torch.stack((torch.diag_embed(keep), adj), 1)


def support_pe_2d(coords):
    """SinePositionalEncoding2D.forward_coordinates, positional_encoding.py:55-82 (scale = 2*3.14159265359)."""
    scale = 2 * 3.14159265359
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * (dim_t // 2) / 128)
    px = (coords[:, :, 0] * scale)[:, :, None] / dim_t
    py = (coords[:, :, 1] * scale)[:, :, None] / dim_t
    n, k, _ = px.shape
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), 3).view(n, k, -1)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), 3).view(n, k, -1)
    return torch.cat((py, px), 2)


def pe_1d(max_len=100, d=256):
    """PositionalEncoding1D, positional_encoding.py:19-26."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


def _left_aligned(valid):
    """torch._nested_tensor_from_mask_left_aligned: every row = valid prefix then padding."""
    v = valid.long()
    return bool(((v[:, 1:] - v[:, :-1]) <= 0).all())


def support_encoder(coords, enc_mask, skeleton, sd, cfg, train=False, grad_mode=True, prefix="support_encoder"):
    """GeometricSupportEncoder.forward.  enc_mask (N,P) True=ignore (the caller has already applied
    the inversion of cape_model.py:120-124).  `grad_mode` mirrors torch.is_grad_enabled() at the call
    site: in eval mode without grad nn.TransformerEncoder takes the nested-tensor fast path when the
    key-padding mask is left-aligned over the whole batch, and then returns 0 at padded positions."""
    N, P, _ = coords.shape
    x = linear(F.relu(linear(coords, sd, prefix + ".coord_mlp.0")), sd, prefix + ".coord_mlp.2")
    x = x + support_pe_2d(coords)
    x = x + sd[prefix + ".sequence_pos_encoding.pe"][:, :P]
    if cfg.use_gcn_preenc:
        adj = adj_from_skeleton(P, skeleton, enc_mask)
        for g in range(cfg.num_gcn_layers):
            w, b = sd[f"{prefix}.gcn_layers.{g}.conv.weight"], sd[f"{prefix}.gcn_layers.{g}.conv.bias"]
            y = F.linear(x, w[:, :, 0], b)              # (N,P,512)  conv1d k=1
            y = y.view(N, P, 2, -1)                     # [n, v, k, c]
            x = F.relu(torch.einsum("nvkc,nkvw->nwc", y, adj))   # graph_utils.py:172
    all_masked = enc_mask.all(1)
    m = enc_mask.clone()
    m[all_masked, 0] = False                            # geometric_support_encoder.py:201-212
    fast = (not train) and (not grad_mode) and _left_aligned(~m)
    p = cfg.dropout
    for i in range(cfg.support_encoder_layers):
        n = f"{prefix}.transformer_encoder.layers.{i}"
        a = mha(x, x, x, sd, n + ".self_attn", cfg.nheads, key_padding_mask=m, p=p, train=train)
        x = layer_norm(x + _drop(a, p, train), sd, n + ".norm1")
        h = linear(_drop(F.relu(linear(x, sd, n + ".linear1")), p, train), sd, n + ".linear2")
        x = layer_norm(x + _drop(h, p, train), sd, n + ".norm2")
    if fast:
        x = x.masked_fill(m[..., None], 0.0)
    if all_masked.any():
        x = x.clone()
        x[all_masked] = 0.0
    return x


# ----------------------------------------------------------------------------
# CAPEModel.forward / forward_inference: models/cape_model.py:79-209
# ----------------------------------------------------------------------------

def cape_forward(sd, cfg, images, support_coords, support_mask, targets, skeleton, train=False, grad_mode=True):
    """Teacher-forced forward.  Returns dict like roomformer_v2.py:350-358."""
    support_mask = support_mask.bool()
    support = support_encoder(support_coords, ~support_mask, skeleton, sd, cfg, train, grad_mode)
    src, pos, mflat, shapes, vr = image_features(images, sd, cfg)
    memory = encoder(src, pos, mflat, shapes, vr, sd, cfg, train)
    L = targets["seq11"].shape[1]
    N = images.shape[0]
    ref0 = torch.sigmoid(sd["base_model.query_embed.weight"])[None, :L].expand(N, -1, -1)
    hs, refs, clss, _ = decoder(seq_embed(sd, targets), ref0, memory, shapes, vr, mflat, support, support_mask,
                                sd, cfg, causal_mask(L), train=train)
    out = {"pred_logits": clss[-1], "pred_coords": refs[-1],
           "pred_room_logits": linear(hs[-1], sd, "base_model.room_class_embed"),
           "aux_outputs": [{"pred_logits": a, "pred_coords": b} for a, b in zip(clss[:-1], refs[:-1])]}
    return out


def next_tokens(cls_type, reg, unfinished, step, cfg, min_len=6):
    """Host token rules of roomformer_v2.py:521-598, vectorised.  cls_type (N,) int, reg (N,2) float32.
    Returns (tok11,tok12,tok21,tok22) int64 (N,), (dx1,dx2,dy1,dy2) float32 (N,), unfinished'."""
    nb = cfg.num_bins
    N = cls_type.shape[0]
    x = torch.minimum(reg[:, 0], torch.ones(()))
    y = torch.minimum(reg[:, 1], torch.ones(()))
    qx, qy = x * (nb - 1), y * (nb - 1)
    fx, fy, cx, cy = torch.floor(qx), torch.floor(qy), torch.ceil(qx), torch.ceil(qy)
    coord = unfinished & ((cls_type == 0) | ((cls_type == 2) & (step < min_len)))
    sep = unfinished & ~coord & (cls_type == 1)
    fin_now = unfinished & ~coord & ~sep              # eos with step >= min_len  (cls==3 cannot occur with 3 classes)
    t = {}
    for name, a, b in (("11", fx, fy), ("12", fx, cy), ("21", cx, fy), ("22", cx, cy)):
        tok = (a * nb + b).long()
        tok = torch.where(coord, tok, torch.full_like(tok, cfg.pad))
        tok = torch.where(sep, torch.full_like(tok, cfg.sep), tok)
        tok = torch.where(fin_now, torch.full_like(tok, cfg.eos), tok)
        t[name] = tok
    dx = torch.where(coord, qx - fx, torch.zeros_like(qx))
    dy = torch.where(coord, qy - fy, torch.zeros_like(qy))
    return t, (dx, 1 - dx, dy, 1 - dy), unfinished & ~fin_now


def stream_from_outputs(logits, coords, cfg, min_len=6):
    """Rebuild the decoder-input token/delta stream (N,T) that an AR run with per-step outputs
    `logits` (N,T,3) / `coords` (N,T,2) fed to itself (roomformer_v2.py:362-383, :521-598)."""
    N, T, _ = logits.shape
    tok = {k: torch.full((N,), cfg.bos, dtype=torch.long) for k in ("11", "12", "21", "22")}
    deltas = (torch.zeros(N), torch.ones(N), torch.zeros(N), torch.ones(N))
    unfinished = torch.ones(N, dtype=torch.bool)
    names = ("seq11", "seq12", "seq21", "seq22", "delta_x1", "delta_x2", "delta_y1", "delta_y2")
    stream = {k: [] for k in names}
    for i in range(T):
        vals = (tok["11"], tok["12"], tok["21"], tok["22"]) + tuple(deltas)
        for k, v in zip(names, vals):
            stream[k].append(v[:, None])
        tok, deltas, unfinished = next_tokens(logits[:, i].argmax(-1), coords[:, i], unfinished, i, cfg, min_len)
    return {k: torch.cat(v, 1) for k, v in stream.items()}


def cape_forward_inference(sd, cfg, images, support_coords, support_mask, skeleton, max_len=None,
                           grad_mode=False, teacher=None):
    """CAPEModel.forward_inference -> RoomFormerV2.forward_inference (cached AR loop),
    roomformer_v2.py:385-677.  If `teacher` (dict of (N,T) token/delta tensors) is given the
    inputs of every step are taken from it instead of the model's own predictions (used to
    compare logits step by step under a fixed token stream)."""
    support_mask = support_mask.bool()
    support = support_encoder(support_coords, ~support_mask, skeleton, sd, cfg, False, grad_mode)
    src, pos, mflat, shapes, vr = image_features(images, sd, cfg)
    memory = encoder(src, pos, mflat, shapes, vr, sd, cfg, False)
    N = images.shape[0]
    max_len = cfg.seq_len if max_len is None else max_len
    ref_all = torch.sigmoid(sd["base_model.query_embed.weight"])[None].expand(N, -1, -1)
    tok = {k: torch.full((N,), cfg.bos, dtype=torch.long) for k in ("11", "12", "21", "22")}
    deltas = (torch.zeros(N), torch.ones(N), torch.zeros(N), torch.ones(N))   # dx1, dx2, dy1, dy2
    unfinished = torch.ones(N, dtype=torch.bool)
    prefix = [torch.zeros(N, 0, cfg.hidden_dim) for _ in range(cfg.dec_layers)]
    logits, coords, hs_all = [], [], []
    stream = {k: [] for k in ("seq11", "seq12", "seq21", "seq22", "delta_x1", "delta_x2", "delta_y1", "delta_y2")}
    i = 0
    while i < max_len and (bool(unfinished.any()) if teacher is None else i < teacher["seq11"].shape[1]):
        if teacher is not None:
            t = {k: teacher[k][:, i:i + 1] for k in stream}
        else:
            t = {"seq11": tok["11"][:, None], "seq12": tok["12"][:, None], "seq21": tok["21"][:, None],
                 "seq22": tok["22"][:, None], "delta_x1": deltas[0][:, None], "delta_x2": deltas[1][:, None],
                 "delta_y1": deltas[2][:, None], "delta_y2": deltas[3][:, None]}
        for k in stream:
            stream[k].append(t[k])
        hs, refs, clss, layer_in = decoder(seq_embed(sd, t), ref_all[:, i:i + 1], memory, shapes, vr, mflat,
                                           support, support_mask, sd, cfg, None, kv_prefix=prefix)
        prefix = [torch.cat([p, li], 1) for p, li in zip(prefix, layer_in)]
        logits.append(clss[-1])
        coords.append(refs[-1])
        hs_all.append(hs[-1])
        tok, deltas, unfinished = next_tokens(clss[-1][:, 0].argmax(-1), refs[-1][:, 0], unfinished, i, cfg)
        i += 1
    pred_logits = torch.cat(logits, 1)
    return {"logits": pred_logits, "coordinates": torch.cat(coords, 1), "sequences": pred_logits.argmax(-1),
            "pred_room_logits": linear(torch.cat(hs_all, 1), sd, "base_model.room_class_embed"),
            "input_stream": {k: torch.cat(v, 1) for k, v in stream.items()}, "unfinished": unfinished}


# ----------------------------------------------------------------------------
# loss: models/cape_losses.py:71-163 + roomformer_v2.py:915-953
# ----------------------------------------------------------------------------

def criterion(outputs, targets, cfg):
    """Returns (loss_dict of 19 entries, weight_dict restricted to those keys, total)."""
    labels = targets["token_labels"]
    vis = targets["visibility_mask"]
    cw = torch.tensor([1.0, 1.0, cfg.eos_weight])

    def one(o):
        m = (labels != -1) & vis
        ce = F.cross_entropy(o["pred_logits"][m], labels[m], weight=cw, reduction="mean")
        mc = (labels == 0) & vis
        l1 = F.l1_loss(o["pred_coords"][mc], targets["target_seq"][mc])
        return ce, l1

    losses = {}
    ce, l1 = one(outputs)
    losses["loss_ce"] = ce
    losses["loss_ce_room"] = torch.tensor(0.0)          # no <cls> labels in CAPE, cape_losses.py:109-117
    losses["loss_coords"] = l1
    losses["cardinality_error"] = 0.0
    for i, a in enumerate(outputs.get("aux_outputs", [])):
        ce, l1 = one(a)
        losses[f"loss_ce_{i}"] = ce
        losses[f"loss_coords_{i}"] = l1
        losses[f"cardinality_error_{i}"] = 0.0
    w = {"loss_ce": cfg.cls_loss_coef, "loss_ce_room": cfg.room_cls_loss_coef, "loss_coords": cfg.coords_loss_coef}
    for i in range(cfg.dec_layers - 1):
        w[f"loss_ce_{i}"] = cfg.cls_loss_coef
        w[f"loss_coords_{i}"] = cfg.coords_loss_coef
    total = sum(losses[k] * w[k] for k in losses if k in w)
    return losses, w, total


# ----------------------------------------------------------------------------
# tokenisation of targets: datasets/mp100_cape.py:625-832, discrete_tokenizer.py
# ----------------------------------------------------------------------------

def tokenize_keypoints(kpts_px, H, W, visibility, cfg, category_id=1):
    """Returns the dict of 13 tensors of length cfg.seq_len for one query instance."""
    nb, L = cfg.num_bins, cfg.seq_len
    K = len(kpts_px)
    if visibility is None:
        visibility = [2] * K
    norm = np.array([[x / W, y / H] for x, y in kpts_px], dtype=np.float64).reshape(K, 2)
    q = np.clip(norm * (nb - 1), 0, nb - 1)
    fl = np.clip(np.floor(q), 0, nb - 1).astype(np.int64)
    ce = np.clip(np.ceil(q), 0, nb - 1).astype(np.int64)

    def seq(a, b):
        body = (a * nb + b).tolist()
        out = [cfg.bos]
        if 1 + len(body) + 1 <= L:                      # tokenizer drops a polygon that does not fit
            out += body
        out += [cfg.pad] * (L - len(out))
        return torch.tensor(out, dtype=torch.long)

    t = {"seq11": seq(fl[:, 0], fl[:, 1]), "seq21": seq(ce[:, 0], fl[:, 1]),
         "seq12": seq(fl[:, 0], ce[:, 1]), "seq22": seq(ce[:, 0], ce[:, 1])}
    target_seq = torch.zeros(L, 2)
    target_seq[:K] = torch.tensor(norm, dtype=torch.float32)
    labels = torch.full((L,), -1, dtype=torch.long)
    labels[:K] = 0
    labels[K] = 2
    mask = torch.zeros(L, dtype=torch.bool)
    mask[:K + 1] = True
    vis = torch.zeros(L, dtype=torch.bool)
    for i in range(K):
        vis[i] = bool(visibility[i] > 0)
    vis[K] = True
    dx1 = torch.zeros(L)
    dy1 = torch.zeros(L)
    d = q - np.floor(q)
    dx1[1:K + 1] = torch.tensor(d[:, 0], dtype=torch.float32)
    dy1[1:K + 1] = torch.tensor(d[:, 1], dtype=torch.float32)
    tpl = torch.full((L,), -1, dtype=torch.long)
    tpl[:K] = category_id
    t.update({"target_seq": target_seq, "token_labels": labels, "mask": mask, "visibility_mask": vis,
              "target_polygon_labels": tpl, "delta_x1": dx1, "delta_x2": 1 - dx1, "delta_y1": dy1,
              "delta_y2": 1 - dy1})
    return t


# ----------------------------------------------------------------------------
# metric: util/eval_utils.py:29-110, util/sequence_utils.py:8-65
# ----------------------------------------------------------------------------

def pck_bbox(pred, gt, bbox_w, bbox_h, visibility=None, threshold=0.2):
    """compute_pck_bbox (util/eval_utils.py:29-110): correct <=> ||pred-gt|| / sqrt(w^2+h^2) < thr,
    counted over keypoints with visibility > 0; returns (pck, num_correct, num_visible)."""
    pred = np.asarray(pred, dtype=np.float64).reshape(-1, 2)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 2)
    vis = np.ones(len(gt), bool) if visibility is None else (np.asarray(visibility) > 0)
    nvis = int(vis.sum())
    if nvis == 0:
        return 0.0, 0, 0
    d = np.sqrt(((pred[vis] - gt[vis]) ** 2).sum(1)) / np.sqrt(bbox_w ** 2 + bbox_h ** 2)
    correct = int((d < threshold).sum())
    return correct / nvis, correct, nvis


def extract_keypoints(pred_coords, token_labels, mask, max_keypoints=None):
    """extract_keypoints_from_sequence (engine_cape.py:304-391): per instance keep tokens with
    mask, then those whose label is <coord>=0, truncate to max_keypoints, zero-pad to batch max."""
    per = []
    for i in range(pred_coords.shape[0]):
        c = pred_coords[i][mask[i]]
        l = token_labels[i][mask[i]]
        k = c[l == 0]
        if max_keypoints is not None and len(k) > max_keypoints:
            k = k[:max_keypoints]
        per.append(k)
    m = max((len(k) for k in per), default=0)
    out = torch.zeros(len(per), m, 2)
    for i, k in enumerate(per):
        out[i, :len(k)] = k
    return out


# ------------------------------------------------------------------------------------------------
# Bidirectional cross-attention blocks (models/bixattn.py:5-235).  Never executed by the reference's CLI (only the
# unreachable decoder layer V3 instantiates them, deformable_transformer_v2.py:894-900): class-level restatement.
# timm.layers.Mlp (requirements_cape.txt:33 `timm>=0.9.0`, not installed offline) is restated from its published
# definition fc1 -> GELU(erf) -> fc2; DropPath / dropout are identities in eval mode.
# ------------------------------------------------------------------------------------------------
def _bix_mlp(x, sd, name):
    return linear(F.gelu(linear(x, sd, name + ".fc1")), sd, name + ".fc2")


def _bix_ls(x, sd, name):
    g = sd.get(name + ".gamma")
    return x if g is None else x * g


def bixattn(x_lat, x_pat, sd, name, heads=8):
    """BiXAttn.forward (bixattn.py:63-85): one similarity, softmax over patches for the latents and over latents for the
    patches."""
    B, Nl, _ = x_lat.shape
    Np = x_pat.shape[1]
    rv_l = linear(x_lat, sd, name + ".rv_latents", bias=(name + ".rv_latents.bias") in sd)
    rv_p = linear(x_pat, sd, name + ".rv_patches", bias=(name + ".rv_patches.bias") in sd)
    D = rv_l.shape[-1] // 2
    hd = D // heads
    split = lambda t, n: t.reshape(B, n, 2, heads, hd).permute(2, 0, 3, 1, 4)
    r_l, v_l = split(rv_l, Nl)
    r_p, v_p = split(rv_p, Np)
    sim = (r_l @ r_p.transpose(-2, -1)) * hd ** -0.5
    a = sim.softmax(-1)
    at = sim.transpose(-2, -1).softmax(-1)
    out_l = linear((a @ v_p).transpose(1, 2).reshape(B, Nl, D), sd, name + ".proj_lat")
    out_p = linear((at @ v_l).transpose(1, 2).reshape(B, Np, D), sd, name + ".proj_pat")
    return out_l, out_p


def bixattn_block(x_lat, x_pat, sd, name, heads=8):
    """BiXAttnBlock.forward (bixattn.py:132-141)."""
    a_l, a_p = bixattn(layer_norm(x_lat, sd, name + ".norm1_lat"), layer_norm(x_pat, sd, name + ".norm1_pat"), sd, name + ".attn", heads)
    x_lat = x_lat + _bix_ls(a_l, sd, name + ".ls1_lat")
    x_lat = x_lat + _bix_ls(_bix_mlp(layer_norm(x_lat, sd, name + ".norm2_lat"), sd, name + ".mlp_lat"), sd, name + ".ls2_lat")
    x_pat = x_pat + _bix_ls(a_p, sd, name + ".ls1_pat")
    x_pat = x_pat + _bix_ls(_bix_mlp(layer_norm(x_pat, sd, name + ".norm2_pat"), sd, name + ".mlp_pat"), sd, name + ".ls2_pat")
    return x_lat, x_pat


def ca_one_sided_block(x_lat, x_pat, sd, name, heads=8):
    """CAOneSidedBlock.forward (bixattn.py:219-235) with CrossAttentionOneSided (:166-181)."""
    B, Nl, _ = x_lat.shape
    Np = x_pat.shape[1]
    xl, xp = layer_norm(x_lat, sd, name + ".norm1_lat"), layer_norm(x_pat, sd, name + ".norm1_pat")
    an = name + ".attn"
    r_l = linear(xl, sd, an + ".r_latents", bias=(an + ".r_latents.bias") in sd)
    rv_p = linear(xp, sd, an + ".rv_patches", bias=(an + ".rv_patches.bias") in sd)
    D = r_l.shape[-1]
    hd = D // heads
    r_l = r_l.reshape(B, Nl, heads, hd).transpose(1, 2)
    r_p, v_p = rv_p.reshape(B, Np, 2, heads, hd).permute(2, 0, 3, 1, 4)
    a = ((r_l @ r_p.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    out = linear((a @ v_p).transpose(1, 2).reshape(B, Nl, D), sd, an + ".proj_lat")
    x_lat = x_lat + _bix_ls(out, sd, name + ".ls1_lat")
    x_lat = x_lat + _bix_ls(_bix_mlp(layer_norm(x_lat, sd, name + ".norm2_lat"), sd, name + ".mlp_lat"), sd, name + ".ls2_lat")
    return x_lat
