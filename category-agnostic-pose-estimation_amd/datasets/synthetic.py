"""Synthetic MP-100-shaped episodes: there is no dataset on the GPU box, so
benchmarks and smoke tests draw seeded episodes with the shapes/statistics of the real loader:
images U[0,1), P keypoints U[0,R) px, 50 % of episodes with two invisible support keypoints, chain
skeleton, category ids cycling 1..10, bbox sides U[64,512)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from .keypoint_tokenization import tokenize_keypoints


class SyntheticEpisodes(Dataset):
    def __init__(self, tokenizer, num_episodes, image_size=256, num_keypoints=17, queries_per_episode=2, num_support=1,
                 seed=0):
        self.tok, self.n, self.R, self.P, self.K, self.S, self.seed = (tokenizer, num_episodes, image_size, num_keypoints,
                                                                      queries_per_episode, num_support, seed)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.Generator(np.random.PCG64([self.seed, idx]))
        R, P = self.R, self.P
        cat = 1 + idx % 10
        sup_c, sup_m = [], []
        for s in range(self.S):
            c = rng.random((P, 2), dtype=np.float32)
            vis = np.full(P, 2)
            if idx % 2 == 1:
                vis[rng.choice(P, size=min(2, P), replace=False)] = 0
            sup_c.append(torch.from_numpy(c)); sup_m.append(torch.from_numpy(vis == 0))
        q_imgs, q_tgts, q_meta = [], [], []
        for k in range(self.K):
            q_imgs.append(torch.from_numpy(rng.random((3, R, R), dtype=np.float32)))
            kp = rng.random((P, 2)) * R
            vis = np.full(P, 2)
            if k % 2 == 1:
                vis[rng.choice(P, size=min(2, P), replace=False)] = 0
            q_tgts.append(tokenize_keypoints(self.tok, [tuple(p) for p in kp], R, R, list(vis), cat))
            q_meta.append({"bbox_width": float(rng.uniform(64, 512)), "bbox_height": float(rng.uniform(64, 512)),
                           "visibility": vis.tolist(), "keypoints": (kp / R).astype(np.float32), "category_id": cat})
        skel = [[i, i + 1] for i in range(P - 1)]
        return {"support_coords": sup_c, "support_masks": sup_m, "support_skeletons": [skel] * self.S,
                "support_images": [None] * self.S, "support_metadata": {"category_id": cat},
                "query_images": q_imgs, "query_targets": q_tgts, "query_metadata": q_meta, "category_id": cat}
