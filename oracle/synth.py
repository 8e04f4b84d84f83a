"""TEST INFRASTRUCTURE ONLY -- seeded synthetic MP-100-shaped episodes.

numpy `Generator(PCG64(seed))` only, so the same inputs are re-created bit-for-bit on any box.
"""
import numpy as np
import torch

from . import cape_ref


def make_batch(seed, B, K, R, P, cfg, n_invisible=(0, 2), tokenizer=None):
    """B episodes x K queries of RxR images with P keypoints.

    n_invisible: per-episode cycle of how many support keypoints get visibility 0
    (0 -> exercises the all-visible => zero-features branch).
    `tokenizer(kpts_px, H, W, vis, category_id)` defaults to the oracle's restatement.
    Returns dict(images (N,3,R,R), support_coords (N,P,2), support_mask (N,P) bool [True = invisible],
    targets {13 x (N,L)}, skeleton list[N], bbox (N,2), visibility (N,P), gt_kpts (N,P,2) in [0,1]).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    tok = tokenizer or (lambda k, h, w, v, c: cape_ref.tokenize_keypoints(k, h, w, v, cfg, c))
    N = B * K
    images = torch.from_numpy(rng.random((N, 3, R, R), dtype=np.float32))
    sup = rng.random((B, P, 2), dtype=np.float32)
    vis_s = np.full((B, P), 2, dtype=np.int64)
    for b in range(B):
        k = n_invisible[b % len(n_invisible)]
        if k:
            vis_s[b, rng.choice(P, size=k, replace=False)] = 0
    kp = rng.random((N, P, 2)) * R
    vis_q = np.full((N, P), 2, dtype=np.int64)
    for n in range(N):
        if n % 2 == 1:
            vis_q[n, rng.choice(P, size=min(2, P), replace=False)] = 0
    per = [tok([tuple(p) for p in kp[n]], R, R, list(vis_q[n]), 1 + (n // K) % 10) for n in range(N)]
    targets = {k: torch.stack([t[k] for t in per]) for k in per[0]}
    skeleton = [[[i, i + 1] for i in range(P - 1)] for _ in range(N)]
    return {
        "images": images,
        "support_coords": torch.from_numpy(sup).repeat_interleave(K, 0),
        "support_mask": torch.from_numpy(vis_s == 0).repeat_interleave(K, 0),
        "targets": targets,
        "skeleton": skeleton,
        "bbox": torch.from_numpy(rng.uniform(64, 512, (N, 2)).astype(np.float32)),
        "visibility": torch.from_numpy(vis_q),
        "gt_kpts": torch.from_numpy((kp / R).astype(np.float32)),
    }
