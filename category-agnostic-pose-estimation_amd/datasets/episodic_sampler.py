"""Episodic batch assembly (reference `datasets/episodic_sampler.py:372-526`): `episodic_collate_fn`
pads supports to the batch maximum (pad mask False), mean-pools K-shot supports, ORs their masks, keeps the
first support's skeleton, repeats everything once per query, stacks the 13 target tensors."""
import torch


def episodic_collate_fn(batch):
    n_sup = len(batch[0]["support_coords"]) if batch else 1
    coords, masks, skels, meta, q_imgs, q_tgts, q_meta, cats, s_imgs = [], [], [], [], [], [], [], [], []
    for ep in batch:
        coords.extend(ep["support_coords"]); masks.extend(ep["support_masks"]); skels.extend(ep["support_skeletons"])
        s_imgs.extend(ep.get("support_images", [None] * n_sup))
        meta.extend([ep.get("support_metadata", {})] * n_sup)
        q_imgs.extend(ep["query_images"]); q_tgts.extend(ep["query_targets"]); q_meta.extend(ep["query_metadata"])
        cats.append(ep["category_id"])
    P = max(c.shape[0] for c in coords)
    pc, pm = [], []
    for c, m in zip(coords, masks):
        pad = P - c.shape[0]
        if pad:
            c = torch.cat([c, torch.zeros(pad, 2)], 0)
            m = torch.cat([m, torch.zeros(pad, dtype=torch.bool)], 0)
        pc.append(c); pm.append(m)
    B = len(batch)
    K = len(q_imgs) // B
    sc = torch.stack(pc).view(B, n_sup, P, 2).mean(1)
    sm = torch.stack(pm).view(B, n_sup, P).any(1)
    support_images = None
    if s_imgs and s_imgs[0] is not None:
        si = torch.stack(s_imgs)
        support_images = si.view(B, n_sup, *si.shape[1:])[:, 0].repeat_interleave(K, 0)
    first_skel = [skels[i * n_sup] for i in range(B)]
    return {
        "support_images": support_images,
        "support_coords": sc.repeat_interleave(K, 0),
        "support_masks": sm.repeat_interleave(K, 0),
        "support_skeletons": [s for s in first_skel for _ in range(K)],
        "support_metadata": [meta[i * n_sup] for i in range(B) for _ in range(K)],
        "query_images": torch.stack(q_imgs),
        "query_targets": {k: torch.stack([t[k] for t in q_tgts]) for k in q_tgts[0]},
        "query_metadata": q_meta,
        "category_ids": torch.tensor(cats, dtype=torch.long).repeat_interleave(K),
    }
