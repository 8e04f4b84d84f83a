"""TEST INFRASTRUCTURE ONLY -- procedural (seeded, reference-free) weights.

SURVEY.md Appendix A step 9: every floating tensor of the 751-entry CAPE `state_dict` is
filled from `numpy.random.Generator(PCG64(crc32(key)))`, scaled by fan-in, so that the same
weights can be re-created on the GPU box from the committed key/shape spec alone
(`tests/golden/state_dict_spec.json`, data emitted by `oracle/make_golden.py`).
"""
import json
import os
import zlib

import numpy as np
import torch

SPEC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "state_dict_spec.json")

# keys that alias the same tensor in the reference (roomformer_v2.py:245-246)
_ALIAS = (("base_model.transformer.decoder.class_embed.", "base_model.class_embed."),
          ("base_model.transformer.decoder.coords_embed.", "base_model.coords_embed."))

# tensors kept as the reference constructs them (deterministic buffers)
# (transformer.pos_embed is NOT one of them: DeformableTransformer._reset_parameters xavier-inits every
# dim>1 parameter, deformable_transformer_v2.py:148-151, so the sincos table is overwritten; it is unused.)
KEEP_AS_BUILT = ("base_model.attention_mask", "support_encoder.sequence_pos_encoding.pe")


def canonical_key(key: str) -> str:
    for a, b in _ALIAS:
        if key.startswith(a):
            return b + key[len(a):]
    return key


def load_spec(path: str = SPEC_PATH):
    with open(path) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def tensor_for(key: str, shape) -> torch.Tensor:
    """Deterministic float32 tensor for a state_dict entry."""
    key = canonical_key(key)
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(key.encode())))
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    is_norm = (".norm" in key or "_norm" in key or ".bn" in key or "downsample.1" in key
               or "layer_norms" in key or (".input_proj." in key and key.rsplit(".", 2)[-2] == "1"))
    if leaf == "running_var" or (is_norm and leaf == "weight"):
        a = rng.uniform(0.5, 1.5, shape)
    elif leaf == "running_mean" or (is_norm and leaf == "bias"):
        a = rng.uniform(-0.1, 0.1, shape)
    elif key.endswith("query_embed.weight"):
        a = rng.uniform(-2.0, 2.0, shape)            # sigmoid -> reference points spread over (0.12, 0.88)
    elif key.endswith("level_embed"):
        a = rng.normal(0.0, 1.0, shape)
    elif key.endswith("token_embed.weight"):
        a = rng.normal(0.0, shape[-1] ** -0.5, shape)
        a[-61:] = a[-61:]                             # rows >= 1940 unused
        a[1939] = 0.0                                # PAD row is zero (padding_idx)
    elif "sampling_offsets.bias" in key:
        a = rng.uniform(-2.0, 2.0, shape)            # offsets of up to two pixels
    elif "class_embed" in key and leaf == "bias":
        a = rng.uniform(-0.5, 0.5, shape)            # distinct per-class biases (argmax margins)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        bound = (3.0 / fan_in) ** 0.5
        if "sampling_offsets.weight" in key:
            bound *= 0.5
        if key.endswith("layers.2.weight") and "coords_embed" in key:
            bound *= 0.25                            # refinement deltas stay moderate
        a = rng.uniform(-bound, bound, shape)
    else:
        a = rng.uniform(-0.05, 0.05, shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def procedural_state_dict(spec=None, built=None):
    """spec: list of (key, shape).  built: optional dict with the as-constructed deterministic buffers
    (attention_mask, pos_embed, pe); when absent they are re-created from their formulas."""
    spec = load_spec() if spec is None else spec
    sd = {}
    for key, shape in spec:
        if key in KEEP_AS_BUILT:
            if built is not None and key in built:
                sd[key] = built[key].clone()
            else:
                sd[key] = _as_built(key, shape)
        else:
            sd[key] = tensor_for(key, shape)
    return sd


def _as_built(key, shape):
    import math
    if key == "base_model.attention_mask":
        L = shape[0]
        return torch.triu(torch.full((L, L), float("-inf")), diagonal=1)
    if key == "support_encoder.sequence_pos_encoding.pe":
        _, L, D = shape
        pe = torch.zeros(L, D)
        position = torch.arange(0, L, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2).float() * (-math.log(10000.0) / D))
        pe[:, 0::2] = torch.sin(position * div)
        pe[:, 1::2] = torch.cos(position * div)
        return pe[None]
    raise KeyError(key)
