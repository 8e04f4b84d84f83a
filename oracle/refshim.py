"""TEST INFRASTRUCTURE ONLY -- import shim for the read-only reference tree.

Makes `/root/reference` importable *in the build container* (torchvision, timm and
pycocotools are not installed there) so that `oracle/make_golden.py` can run the
real reference and emit golden vectors, and so that CPU tests can cross-check the
restatement in `oracle/cape_ref.py` against it.  Nothing here is shipped, nothing
here is measured, and nothing here is available on the GPU box (the reference does
not travel).  Only `tests/`, `oracle/make_golden.py` may import this module.

Recipe follows SURVEY.md Appendix A:
  * torchvision stub: `resnet50` = a locally written ResNet-50 v1.5 *architecture*
    (torchvision's published layout: Bottleneck 1x1 -> 3x3(stride) -> 1x1x4, bias-free
    convs, downsample 1x1(stride)+norm; child names conv1,bn1,relu,maxpool,layer1..4,
    avgpool,fc).  `weights=` is ignored (no download is possible or attempted).
  * timm.layers stub: DropPath (identity in eval) and Mlp (fc1 -> act -> fc2).
  * pycocotools.coco stub: dummy COCO class.
"""
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("CAPE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _ResNet50(nn.Module):
    def __init__(self, norm_layer):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, 3, 1, norm_layer)
        self.layer2 = self._make(128, 4, 2, norm_layer)
        self.layer3 = self._make(256, 6, 2, norm_layer)
        self.layer4 = self._make(512, 3, 2, norm_layer)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride, norm_layer):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               norm_layer(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, ds, norm_layer)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, 1, None, norm_layer))
        return nn.Sequential(*layers)


def _resnet50(replace_stride_with_dilation=None, weights=None, norm_layer=None, **kw):
    assert not any(replace_stride_with_dilation or [False]), "dilation unsupported in shim"
    return _ResNet50(norm_layer or nn.BatchNorm2d)


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m / keep


class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **kw):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


_installed = False


def install():
    """Register the stub modules and put the reference root first on sys.path."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present (this shim only works in the build container)")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = mod("torchvision", __version__="0.25.0", __path__=[])
    ops = mod("torchvision.ops", __path__=[])
    misc = mod("torchvision.ops.misc", interpolate=torch.nn.functional.interpolate)
    models = mod("torchvision.models", resnet50=_resnet50, __path__=[])
    mutils = mod("torchvision.models._utils", IntermediateLayerGetter=_IntermediateLayerGetter)
    tv.ops, tv.models, ops.misc, models._utils = ops, models, misc, mutils
    timm = mod("timm", __path__=[])
    tl = mod("timm.layers", DropPath=_DropPath, Mlp=_Mlp)
    timm.layers = tl
    pc = mod("pycocotools", __path__=[])
    coco = mod("pycocotools.coco", COCO=type("COCO", (), {}))
    pc.coco = coco
    # the reference's own top-level packages are called models/datasets/util: make sure ours
    # (if any were imported under those names) do not shadow them
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "datasets", "util")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def build_reference(extra_args=(), seq_len=None):
    """Build (args, tokenizer, CAPEModel, criterion) of the reference on CPU (Appendix A steps 2-4)."""
    install()
    import argparse
    import math
    from models.train_cape_episodic import get_args_parser
    from models import build_model
    from models.cape_model import build_cape_model
    from models.cape_losses import build_cape_criterion
    from datasets.discrete_tokenizer import DiscreteTokenizerV2

    argv = ["--use_geometric_encoder", "--use_gcn_preenc", "--device", "cpu"] + list(extra_args)
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(argv)
    tok = DiscreteTokenizerV2(num_bins=int(math.sqrt(args.vocab_size)), seq_len=args.seq_len, add_cls=False)
    base, _ = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base)
    crit = build_cape_criterion(args, num_classes=3)
    return args, tok, model, crit


def reference_tokenize(tok, keypoints_px, H, W, visibility, category_id=1):
    """Call the reference tokeniser unbound (Appendix A step 5)."""
    install()
    from datasets.mp100_cape import MP100CAPE
    ns = types.SimpleNamespace(tokenizer=tok, _current_category_id=category_id)
    return MP100CAPE._tokenize_keypoints(ns, keypoints_px, H, W, visibility)
