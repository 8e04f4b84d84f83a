"""Episodic sampling and batch assembly (reference `datasets/episodic_sampler.py`): `EpisodicSampler` (:13-119: categories of
a split with enough examples, one episode = K support + Q query indices of one category), `EpisodicDataset` (:120-370:
episode -> support graphs normalised by the crop size and clamped, mask True = invisible, query images / targets /
metadata, resampling on missing images, optional fixed validation episodes), `episodic_collate_fn` (:372-482: pads supports
to the batch maximum (pad mask False), mean-pools K-shot supports, ORs their masks, keeps the first support's skeleton,
repeats everything once per query, stacks the 13 target tensors), `build_episodic_dataloader` (:483-525)."""
import json
import random
from collections import defaultdict

import numpy as np
import torch
import torch.utils.data as data


class EpisodicSampler:
    def __init__(self, dataset, category_split_file, split="train", num_queries_per_episode=2, num_support_per_episode=1, seed=None):
        self.dataset, self.split = dataset, split
        self.num_queries, self.num_support = num_queries_per_episode, num_support_per_episode
        if seed is not None:
            random.seed(seed)
            np.random.seed(seed)
        with open(category_split_file) as f:
            splits = json.load(f)
        if split not in ("train", "val", "test"):
            raise ValueError(f"Unknown split: {split}. Must be 'train', 'val', or 'test'")
        self.categories = splits[split]
        original = list(self.categories)
        self.category_to_indices = defaultdict(list)
        for idx in range(len(dataset)):
            try:
                anns = dataset.coco.loadAnns(dataset.coco.getAnnIds(imgIds=dataset.ids[idx]))
                if len(anns) > 0:
                    cat = anns[0].get("category_id", 0)
                    if cat in self.categories:
                        self.category_to_indices[cat].append(idx)
            except Exception:
                continue
        need = num_queries_per_episode + num_support_per_episode
        self.categories = [c for c in self.categories if len(self.category_to_indices[c]) >= need]
        if len(self.categories) == 0:
            counts = {c: len(self.category_to_indices[c]) for c in original}
            raise ValueError(f"No valid categories found for {split} split after filtering: at least {need} examples per "
                             f"category are required (num_support={num_support_per_episode} + num_queries="
                             f"{num_queries_per_episode}); samples per category: {counts}")

    def sample_episode(self):
        cat = random.choice(self.categories)
        picked = random.sample(self.category_to_indices[cat], self.num_queries + self.num_support)
        return {"category_id": cat, "support_indices": picked[:self.num_support], "query_indices": picked[self.num_support:]}

    def __len__(self):
        return sum(len(v) for v in self.category_to_indices.values()) // self.num_queries


class EpisodicDataset(data.Dataset):
    def __init__(self, base_dataset, category_split_file, split="train", num_queries_per_episode=2, num_support_per_episode=1,
                 episodes_per_epoch=1000, seed=None, fixed_episodes=False, load_support_images=True):
        self.base_dataset, self.episodes_per_epoch = base_dataset, episodes_per_epoch
        self.num_support, self.fixed_episodes, self.load_support_images = num_support_per_episode, fixed_episodes, load_support_images
        self._cached_episodes = None
        self.sampler = EpisodicSampler(base_dataset, category_split_file, split=split, num_queries_per_episode=num_queries_per_episode,
                                       num_support_per_episode=num_support_per_episode, seed=seed)
        if fixed_episodes:
            self._cached_episodes = [self.sampler.sample_episode() for _ in range(episodes_per_epoch)]

    def __len__(self):
        return self.episodes_per_epoch

    def _support(self, d):
        c = torch.tensor(d["keypoints"], dtype=torch.float32)
        c[:, 0] /= d["width"]
        c[:, 1] /= d["height"]
        c = c.clamp(0.0, 1.0)
        if "visibility" not in d:
            raise KeyError(f"Support data for image {d.get('image_id', 'unknown')} is missing 'visibility' field.")
        if len(d["visibility"]) != len(c):
            raise ValueError(f"Support visibility length ({len(d['visibility'])}) doesn't match keypoints length ({len(c)})")
        return c, torch.tensor([v == 0 for v in d["visibility"]], dtype=torch.bool), d.get("skeleton", [])

    def __getitem__(self, idx):
        from .mp100_cape import ImageNotFoundError
        use_fixed = self.fixed_episodes and self._cached_episodes is not None
        for attempt in range(100):
            try:
                ep = self._cached_episodes[idx % len(self._cached_episodes)] if (use_fixed and attempt == 0) else self.sampler.sample_episode()
                sup = [self.base_dataset[i] for i in ep["support_indices"]]
                parts = [self._support(d) for d in sup]
                q_imgs, q_tgts, q_meta = [], [], []
                for qi in ep["query_indices"]:
                    q = self.base_dataset[qi]
                    if "visibility" not in q:
                        raise KeyError(f"Query data for image {q.get('image_id', 'unknown')} is missing 'visibility' field.")
                    if len(q["visibility"]) != len(q["keypoints"]):
                        raise ValueError(f"Visibility length ({len(q['visibility'])}) doesn't match keypoints length ({len(q['keypoints'])})")
                    q_imgs.append(q["image"] if q.get("image") is not None else (q["raw_crop"], q["plan"]))
                    q_tgts.append(q["seq_data"])
                    q_meta.append({"image_id": q["image_id"], "height": q["height"], "width": q["width"], "keypoints": q["keypoints"],
                                   "num_keypoints": q["num_keypoints"],
                                   "num_visible_keypoints": q.get("num_visible_keypoints", q["num_keypoints"]),
                                   "bbox": q.get("bbox", [0, 0, q["width"], q["height"]]), "bbox_width": q.get("bbox_width", q["width"]),
                                   "bbox_height": q.get("bbox_height", q["height"]), "visibility": q["visibility"]})
                first = sup[0]
                return {"support_images": [d["image"] for d in sup] if self.load_support_images else [None] * len(sup),
                        "support_coords": [p[0] for p in parts], "support_masks": [p[1] for p in parts],
                        "support_skeletons": [p[2] for p in parts],
                        "support_metadata": {"image_id": first.get("image_id"), "category_id": first.get("category_id"),
                                             "bbox_width": first.get("bbox_width", first.get("width")),
                                             "bbox_height": first.get("bbox_height", first.get("height"))},
                        "query_images": q_imgs, "query_targets": q_tgts, "query_metadata": q_meta, "category_id": ep["category_id"]}
            except ImageNotFoundError:
                continue
        raise RuntimeError("Failed to find valid episode after 100 attempts. This may indicate too many missing images in the dataset.")


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
    deferred = bool(q_imgs) and isinstance(q_imgs[0], tuple)
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
        # deferred pixels (MP100CAPE(defer_pixels=True)): raw uint8 crops + plans travel, transforms.DeviceImagePipeline makes the batch
        "query_images": torch.stack(q_imgs) if not deferred else None,
        **({"query_raw": q_imgs} if deferred else {}),
        "query_targets": {k: torch.stack([t[k] for t in q_tgts]) for k in q_tgts[0]},
        "query_metadata": q_meta,
        "category_ids": torch.tensor(cats, dtype=torch.long).repeat_interleave(K),
    }


def build_episodic_dataloader(base_dataset, category_split_file, split="train", batch_size=2, num_queries_per_episode=2,
                              num_support_per_episode=1, episodes_per_epoch=1000, num_workers=16, seed=None, fixed_episodes=False,
                              load_support_images=True):
    ds = EpisodicDataset(base_dataset, category_split_file, split=split, num_queries_per_episode=num_queries_per_episode,
                         num_support_per_episode=num_support_per_episode, episodes_per_epoch=episodes_per_epoch, seed=seed,
                         fixed_episodes=fixed_episodes, load_support_images=load_support_images)
    return data.DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=num_workers, collate_fn=episodic_collate_fn,
                           pin_memory=True)
