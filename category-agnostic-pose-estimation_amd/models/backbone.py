"""ResNet-50 trunk on MI355X: NHWC activations, channels_last weights, every convolution an
implicit GEMM on fp32 MFMA with FrozenBatchNorm folded into the epilogue (scale/shift), ReLU and the
residual add fused.  Mirrors the module tree / parameter names of the reference
(`models/backbone.py:13-103` + torchvision ResNet-50 v1.5) so `state_dict`s interchange:
`backbone.0.body.{conv1,bn1,layerK.B.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}}`.

The `IMAGENET1K_V1` checkpoint of the reference (`backbone.py:75-78`) cannot be fetched offline; the
trunk is initialised with torchvision's from-scratch scheme and real weights arrive through
`load_state_dict` (same keys).
"""
import math

import torch
from torch import nn

from ..hip import functional as HF
from ..hip import ops
from ..util.misc import NestedTensor, cached_zero_mask
from .position_encoding import build_position_encoding


class Conv2dCL(nn.Module):
    """Parameter holder for a bias-free/biased conv whose weight (O, C, KH, KW) is stored channels_last
    (physical (O, KH, KW, C)): the layout the implicit-GEMM kernels stream with 16-byte loads."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.stride, self.padding, self.kernel_size = stride, padding, k

    def _load_from_state_dict(self, state_dict, prefix, *a, **kw):
        super()._load_from_state_dict(state_dict, prefix, *a, **kw)
        # copy_ keeps our strides; make sure nothing replaced the storage layout
        assert self.weight.permute(0, 2, 3, 1).is_contiguous()


class FrozenBatchNorm2d(nn.Module):
    """Fixed statistics and affine (reference `backbone.py:13-40`, eps 1e-5).  Never applied on its own:
    `folded()` gives the per-channel (scale, shift) consumed by the conv epilogue."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps
        self._cache = None

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self._cache = None

    def folded(self):
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.data_ptr())
        if self._cache is None or self._cache[0] != key:
            with torch.no_grad():
                self._cache = (key, ops.bn_fold(self.weight, self.bias, self.running_mean, self.running_var, self.eps))
        return self._cache[1]


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2dCL(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = Conv2dCL(planes, planes, 3, stride=stride, padding=1)      # v1.5: stride on the 3x3
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = Conv2dCL(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        bnd = self.downsample[1].folded() if self.downsample is not None else None
        wd = self.downsample[0].weight if self.downsample is not None else None
        # one autograd node per block (HF.BottleneckFn): relu(bn3(conv3(...)) + shortcut) in one epilogue, and in the backward
        # conv1's data gradient accumulates into the shortcut's gradient
        return HF.bottleneck(x, self.conv1.weight, self.conv2.weight, self.conv3.weight, wd, self.bn1.folded(), self.bn2.folded(),
                             self.bn3.folded(), bnd, self.stride)


class ResNet50Body(nn.Module):
    """conv1/bn1/maxpool/layer1..4 with the child names IntermediateLayerGetter exposes."""

    def __init__(self, input_channels=3):
        super().__init__()
        self.input_channels = input_channels
        self.conv1 = Conv2dCL(input_channels, 64, 7, stride=2, padding=3)
        # the reference replaces conv1 by a default-initialised nn.Conv2d (backbone.py:79)
        w = torch.empty(64, input_channels, 7, 7)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        with torch.no_grad():
            self.conv1.weight.copy_(w)
        self.bn1 = FrozenBatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        self._stem_cache = None

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(Conv2dCL(self.inplanes, planes * 4, 1, stride=stride), FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _stem_weight(self):
        """7x7 stem weight padded to 4 input channels, physical (64, 7, 7, 4) (frozen -> cached)."""
        w = self.conv1.weight
        key = (w._version, w.data_ptr())
        if self._stem_cache is None or self._stem_cache[0] != key:
            with torch.no_grad():
                cp = (self.input_channels + 3) // 4 * 4
                wp = torch.zeros(64, 7, 7, cp, dtype=torch.float32, device=w.device)
                wp[..., : self.input_channels] = w.permute(0, 2, 3, 1)
                self._stem_cache = (key, wp.permute(0, 3, 1, 2))
        return self._stem_cache[1]

    def forward(self, images_nchw):
        """(N, C, H, W) -> [C3, C4, C5] as NHWC tensors."""
        with torch.no_grad():       # stem and layer1 are frozen (backbone.py:44-46): no graph is needed
            cp = (self.input_channels + 3) // 4 * 4
            x = ops.nchw_to_nhwc(images_nchw.contiguous(), cp)
            s, b = self.bn1.folded()
            x = HF.conv_bn_act(x, self._stem_weight(), s, b, 2, 3, relu=True)
            x = ops.maxpool3x3s2(x)
            frozen1 = not any(p.requires_grad for p in self.layer1.parameters())
            if frozen1:
                x = self.layer1(x)
        if not frozen1:
            x = self.layer1(x)
        c3, c3n = HF.fanout(self.layer2(x), 2)    # each level also feeds the next stage
        c4, c4n = HF.fanout(self.layer3(c3n), 2)
        c5 = self.layer4(c4n)
        return [c3, c4, c5]


class BackboneBase(nn.Module):
    def __init__(self, body: nn.Module, train_backbone: bool, return_interm_layers: bool):
        super().__init__()
        for name, p in body.named_parameters():
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                p.requires_grad_(False)
        if return_interm_layers:
            self.strides = [8, 16, 32]
            self.num_channels = [512, 1024, 2048]
        else:
            self.strides = [32]
            self.num_channels = [2048]
        self.return_interm_layers = return_interm_layers
        self.body = body

    def forward(self, tensor_list: NestedTensor):
        feats = self.body(tensor_list.tensors)
        if not self.return_interm_layers:
            feats = feats[-1:]
        out = {}
        m = tensor_list.mask
        if getattr(tensor_list, "no_padding", False):       # equally sized images: the resampled mask is all-False at every level
            for i, x in enumerate(feats):
                out[str(i)] = NestedTensor(x, cached_zero_mask(x.shape[0], x.shape[1], x.shape[2], x.device))
            return out
        for i, x in enumerate(feats):
            # nearest-neighbour mask resize (F.interpolate default): index floor(i * H / h)
            h, w = x.shape[1], x.shape[2]
            iy = (torch.arange(h, device=m.device) * m.shape[1] // h)
            ix = (torch.arange(w, device=m.device) * m.shape[2] // w)
            out[str(i)] = NestedTensor(x, m[:, iy][:, :, ix])
        return out


class Backbone(BackboneBase):
    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool, input_channels=1):
        if name != "resnet50":
            raise ValueError(f"cape_amd implements the reference's default backbone resnet50, got {name}")
        if dilation:
            raise ValueError("dilation is not supported by the MI355X backbone")
        super().__init__(ResNet50Body(input_channels), train_backbone, return_interm_layers)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        return [xs[k] for k in sorted(xs)]


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    return_interm_layers = args.num_feature_levels > 1
    backbone = Backbone(args.backbone, train_backbone, return_interm_layers, args.dilation,
                        input_channels=args.input_channels)
    return Joiner(backbone, position_embedding)
