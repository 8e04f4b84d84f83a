"""Bidirectional cross-attention blocks (reference `models/bixattn.py:5-235`) on the MI355X kernels.

The reference never executes these classes (only the unreachable decoder-layer variant V3 builds them,
`deformable_transformer_v2.py:894-900`), so they are provided as standalone ops with the reference's class names,
constructor arguments and `state_dict` keys, forward AND backward (autograd nodes of hip/functional.py: LayerNorm, Linear,
exact GELU, LayerScale residual, the two attention directions as one node), parity at class level against the reference's own
classes: outputs (tests/golden/bixattn.npz) and input / parameter gradients (tests/golden/bixattn_grads.npz).  Training mode is
accepted while every drop rate is 0 (the fixtures' setting); a non-zero rate in training mode raises -- no dropout kernel is
wired into these blocks.

One similarity matrix serves both directions in the reference; here each direction is one pass of the tiled
online-softmax attention kernel (`cape_attn_fwd`, keys staged 256 rows at a time): latents attend over the patches,
patches attend over the latents -- the two softmaxes normalise over different axes, so nothing but the r.r^T products
would be shared, and recomputing them costs less than materialising the (heads, N_lat, N_pat) matrix in HBM."""
import torch
import torch.nn as nn

from ..hip import functional as HF
from ..hip import ops


def _no_active_dropout(mod):
    """Training mode is fine while nothing would drop: the MI355X blocks carry no dropout / drop-path kernel."""
    if not mod.training:
        return
    for m in mod.modules():
        if isinstance(m, nn.Dropout) and m.p > 0:
            raise RuntimeError(f"{type(mod).__name__}: dropout p={m.p} in training mode is not implemented on the MI355X path "
                               "(the reference never trains these blocks; use rate 0 or .eval())")
    if getattr(mod, "drop_path", 0.0):
        raise RuntimeError(f"{type(mod).__name__}: drop_path={mod.drop_path} in training mode is not implemented on the MI355X path")


def _ln(x, norm):
    return HF.add_layernorm(x, None, norm.weight, norm.bias)


def _lin(x, lin):
    return HF.linear(x, lin.weight, lin.bias)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        y = HF.scale_residual(None, x, self.gamma)
        if self.inplace:                                  # x.mul_(gamma) of the reference: only meaningful outside autograd
            with torch.no_grad():
                x.copy_(y)
            return x
        return y


class Mlp(nn.Module):
    """timm.layers.Mlp (fc1 -> GELU -> drop -> fc2 -> drop); only `act_layer=nn.GELU` is implemented."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if act_layer is not nn.GELU:
            raise ValueError("MI355X Mlp: act_layer must be nn.GELU (exact erf form)")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        _no_active_dropout(self)
        return _lin(HF.gelu(_lin(x, self.fc1)), self.fc2)


class _AttnBase(nn.Module):
    def _check(self, dim_attn, num_heads):
        assert dim_attn % num_heads == 0, 'dim_attn MUST be divisible by num_heads'
        if dim_attn // num_heads != 32:
            raise ValueError("MI355X attention kernel: head dimension must be 32 (dim_attn / num_heads)")
        self.num_heads, self.dim_attn = num_heads, dim_attn
        self.scale = (dim_attn // num_heads) ** -0.5



class BiXAttn(_AttnBase):
    def __init__(self, dim_lat, dim_pat, dim_attn, num_heads=8, rv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self._check(dim_attn, num_heads)
        self.rv_latents = nn.Linear(dim_lat, dim_attn * 2, bias=rv_bias)
        self.rv_patches = nn.Linear(dim_pat, dim_attn * 2, bias=rv_bias)
        self.attn_drop, self.attn_dropT = nn.Dropout(attn_drop), nn.Dropout(attn_drop)
        self.proj_lat, self.proj_drop_lat = nn.Linear(dim_attn, dim_lat), nn.Dropout(proj_drop)
        self.proj_pat, self.proj_drop_pat = nn.Linear(dim_attn, dim_pat), nn.Dropout(proj_drop)

    def forward(self, x_latents, x_patches):
        _no_active_dropout(self)
        rv_l, rv_p = _lin(x_latents, self.rv_latents), _lin(x_patches, self.rv_patches)     # (B, N, 2D): [r | v]
        # lat: softmax over the patches; pat: softmax over the latents (one autograd node for both directions)
        lat, pat = HF.bi_attn_core(rv_l, rv_p, self.num_heads, self.scale)
        return _lin(lat, self.proj_lat), _lin(pat, self.proj_pat)


class CrossAttentionOneSided(_AttnBase):
    def __init__(self, dim_lat, dim_pat, dim_attn, num_heads=8, rv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self._check(dim_attn, num_heads)
        self.r_latents = nn.Linear(dim_lat, dim_attn, bias=rv_bias)
        self.rv_patches = nn.Linear(dim_pat, dim_attn * 2, bias=rv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_lat, self.proj_drop_lat = nn.Linear(dim_attn, dim_lat), nn.Dropout(proj_drop)

    def forward(self, x_latents, x_patches):
        _no_active_dropout(self)
        r_l, rv_p = _lin(x_latents, self.r_latents), _lin(x_patches, self.rv_patches)
        return _lin(HF.attn_kv(r_l, rv_p, self.num_heads, self.scale), self.proj_lat)


def _ls(dim, init_values):
    return LayerScale(dim, init_values=init_values) if init_values else nn.Identity()


def _res(x, y, ls):
    return HF.scale_residual(x, y, ls.gamma if isinstance(ls, LayerScale) else None)


class BiXAttnBlock(nn.Module):
    def __init__(self, dim_lat, dim_pat, dim_attn, num_heads, rv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, lat_mlp_ratio=4., pat_mlp_ratio=4.):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise ValueError("MI355X BiXAttnBlock: norm_layer must be nn.LayerNorm")
        self.norm1_lat, self.norm1_pat = norm_layer(dim_lat), norm_layer(dim_pat)
        self.attn = BiXAttn(dim_lat=dim_lat, dim_pat=dim_pat, dim_attn=dim_attn, num_heads=num_heads, rv_bias=rv_bias,
                            attn_drop=attn_drop, proj_drop=drop)
        self.ls1_lat, self.ls1_pat = _ls(dim_lat, init_values), _ls(dim_pat, init_values)
        self.drop_path1_lat = self.drop_path1_pat = self.drop_path2_lat = self.drop_path2_pat = nn.Identity()
        self.drop_path = drop_path
        self.norm2_lat = norm_layer(dim_lat)
        self.mlp_lat = Mlp(in_features=dim_lat, hidden_features=int(dim_lat * lat_mlp_ratio), act_layer=act_layer, drop=drop)
        self.ls2_lat = _ls(dim_lat, init_values)
        self.norm2_pat = norm_layer(dim_pat)
        self.mlp_pat = Mlp(in_features=dim_pat, hidden_features=int(dim_pat * pat_mlp_ratio), act_layer=act_layer, drop=drop)
        self.ls2_pat = _ls(dim_pat, init_values)

    def forward(self, x_latents, x_patches):
        _no_active_dropout(self)
        a_l, a_p = self.attn(_ln(x_latents, self.norm1_lat), _ln(x_patches, self.norm1_pat))
        x_latents = _res(x_latents, a_l, self.ls1_lat)
        x_latents = _res(x_latents, self.mlp_lat(_ln(x_latents, self.norm2_lat)), self.ls2_lat)
        x_patches = _res(x_patches, a_p, self.ls1_pat)
        x_patches = _res(x_patches, self.mlp_pat(_ln(x_patches, self.norm2_pat)), self.ls2_pat)
        return x_latents, x_patches


class CAOneSidedBlock(nn.Module):
    def __init__(self, dim_lat, dim_pat, dim_attn, num_heads, rv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, lat_mlp_ratio=4.):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise ValueError("MI355X CAOneSidedBlock: norm_layer must be nn.LayerNorm")
        self.norm1_lat, self.norm1_pat = norm_layer(dim_lat), norm_layer(dim_pat)
        self.attn = CrossAttentionOneSided(dim_lat=dim_lat, dim_pat=dim_pat, dim_attn=dim_attn, num_heads=num_heads,
                                           rv_bias=rv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.ls1_lat = _ls(dim_lat, init_values)
        self.drop_path1_lat = self.drop_path2_lat = nn.Identity()
        self.drop_path = drop_path
        self.norm2_lat = norm_layer(dim_lat)
        self.mlp_lat = Mlp(in_features=dim_lat, hidden_features=int(dim_lat * lat_mlp_ratio), act_layer=act_layer, drop=drop)
        self.ls2_lat = _ls(dim_lat, init_values)

    def forward(self, x_latents, x_patches):
        _no_active_dropout(self)
        a_l = self.attn(_ln(x_latents, self.norm1_lat), _ln(x_patches, self.norm1_pat))
        x_latents = _res(x_latents, a_l, self.ls1_lat)
        x_latents = _res(x_latents, self.mlp_lat(_ln(x_latents, self.norm2_lat)), self.ls2_lat)
        return x_latents, None
