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


def make_episode(seed, R, P, K, S, cfg, tokenizer=None, category_id=None, n_invisible=None):
    """One episode in the dataset's dict format (what `episodic_collate_fn` consumes, datasets/episodic_sampler.py:372-482):
    S support graphs of P keypoints, K query images RxR with tokenised targets and metadata (bbox, visibility)."""
    rng = np.random.Generator(np.random.PCG64([seed, 7]))
    tok = tokenizer or (lambda k, h, w, v, c: cape_ref.tokenize_keypoints(k, h, w, v, cfg, c))
    cat = int(category_id if category_id is not None else 1 + seed % 10)
    n_inv = (2 if seed % 2 else 0) if n_invisible is None else n_invisible
    sup_c, sup_m = [], []
    for _ in range(S):
        c = rng.random((P, 2), dtype=np.float32)
        vis = np.full(P, 2)
        if n_inv:
            vis[rng.choice(P, size=min(n_inv, P), replace=False)] = 0
        sup_c.append(torch.from_numpy(c))
        sup_m.append(torch.from_numpy(vis == 0))
    q_imgs, q_tgts, q_meta = [], [], []
    for k in range(K):
        q_imgs.append(torch.from_numpy(rng.random((3, R, R), dtype=np.float32)))
        kp = rng.random((P, 2)) * R
        vis = np.full(P, 2)
        if k % 2 == 1:
            vis[rng.choice(P, size=min(2, P), replace=False)] = 0
        q_tgts.append(tok([tuple(p) for p in kp], R, R, list(vis), cat))
        q_meta.append({"bbox_width": float(rng.uniform(64, 512)), "bbox_height": float(rng.uniform(64, 512)),
                       "visibility": [int(v) for v in vis], "category_id": cat})
    skel = [[i, i + 1] for i in range(P - 1)]
    return {"support_coords": sup_c, "support_masks": sup_m, "support_skeletons": [skel] * S,
            "support_images": [None] * S, "support_metadata": {"category_id": cat},
            "query_images": q_imgs, "query_targets": q_tgts, "query_metadata": q_meta, "category_id": cat}
