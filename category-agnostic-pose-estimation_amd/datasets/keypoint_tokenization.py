"""Input contract of the hot path: keypoints -> the 13 target tensors of one query instance
(`MP100CAPE._tokenize_keypoints`, reference `datasets/mp100_cape.py:625-832`).  Host code (data loader
workers); vectorised with numpy."""
import numpy as np
import torch

from .token_types import TokenType


def tokenize_keypoints(tokenizer, keypoints, height, width, visibility=None, category_id=0):
    nb, L = tokenizer.num_bins, tokenizer.seq_len
    K = len(keypoints)
    if visibility is None:
        visibility = [2] * K
    norm = np.asarray([[x / width, y / height] for x, y in keypoints], dtype=np.float64).reshape(K, 2)
    quant = np.clip(norm * (nb - 1), 0, nb - 1)
    lo = np.clip(np.floor(quant), 0, nb - 1).astype(np.int64)
    hi = np.clip(np.ceil(quant), 0, nb - 1).astype(np.int64)

    def ids(ax, ay):
        return [(ax * nb + ay).tolist()]

    seq11 = tokenizer(ids(lo[:, 0], lo[:, 1]), add_bos=True, add_eos=False, dtype=torch.long)
    seq21 = tokenizer(ids(hi[:, 0], lo[:, 1]), add_bos=True, add_eos=False, dtype=torch.long)
    seq12 = tokenizer(ids(lo[:, 0], hi[:, 1]), add_bos=True, add_eos=False, dtype=torch.long)
    seq22 = tokenizer(ids(hi[:, 0], hi[:, 1]), add_bos=True, add_eos=False, dtype=torch.long)

    labels = [TokenType.coord.value] * K + [TokenType.eos.value]
    target = [list(p) for p in norm] + [[0, 0]]
    mask = torch.zeros(L, dtype=torch.bool)
    mask[:len(labels)] = True
    vis_mask = torch.zeros(L, dtype=torch.bool)
    for i in range(min(K, L)):
        vis_mask[i] = bool(visibility[i] > 0)
    if K < L:
        vis_mask[K] = True                                   # the EOS token takes part in the loss
    target_seq = tokenizer._padding(target, [0, 0], dtype=torch.float32)
    token_labels = tokenizer._padding(labels, -1, dtype=torch.long)
    frac = quant - np.floor(quant)
    dx1 = tokenizer._padding([0.0] + frac[:, 0].tolist(), 0, dtype=torch.float32)
    dy1 = tokenizer._padding([0.0] + frac[:, 1].tolist(), 0, dtype=torch.float32)
    tpl = torch.full((L,), -1, dtype=torch.long)
    tpl[:min(K, L)] = category_id
    return {"seq11": seq11, "seq21": seq21, "seq12": seq12, "seq22": seq22, "target_seq": target_seq,
            "token_labels": token_labels, "mask": mask, "visibility_mask": vis_mask, "target_polygon_labels": tpl,
            "delta_x1": dx1, "delta_x2": 1 - dx1, "delta_y1": dy1, "delta_y2": 1 - dy1}
