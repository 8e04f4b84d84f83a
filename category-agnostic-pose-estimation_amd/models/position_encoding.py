"""Image position embedding holder (reference `models/position_encoding.py:7-81`).  The sine embedding has
no parameters; on MI355X it is produced, together with the level embedding, by one kernel that writes
straight into the flattened (N, S, 256) token layout (`hip.functional.level_pos`).  This module only
keeps the configuration and the place in the module tree (`backbone.1`)."""
import math

from torch import nn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if not normalize or temperature != 10000 or (scale is not None and abs(scale - 2 * math.pi) > 1e-9):
            raise ValueError("the MI355X kernel implements the reference's configuration: normalize=True, "
                             "temperature=10000, scale=2*pi")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi


def build_position_encoding(args):
    if args.position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(args.hidden_dim // 2, normalize=True)
    raise ValueError(f"not supported {args.position_embedding} (the reference's learned embedding is not on the CAPE path)")
