"""Host-side mirror of the reference's `models` package for the CAPE hot path
(models/__init__.py:9-27): `build_model(args, train=True, tokenizer=None)`."""
from .roomformer_v2 import build as build_v2


def build_model(args, train=True, tokenizer=None):
    # dropout stream ids number the call sites of ONE model in construction order: a second model built in the same process
    # (a resumed run, an evaluation model) draws the same masks as the first for the same (seed, step)
    from ..hip import ops
    ops._stream_counter[0] = 0
    if not getattr(args, "poly2seq", True):
        return build_v2(args, train)
    # the base model is always built with cape_mode=False; CAPEModel injects the support features
    return build_v2(args, train, tokenizer=tokenizer, cape_mode=False)
