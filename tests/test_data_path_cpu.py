"""Host data path: the MP-100 loader, the episodic sampler / dataset and the transform plans, on a small
COCO-style dataset written to a temp dir (PNG files + annotation JSON + category_splits.json).  Cross-checked against the
reference's own `MP100CAPE` / `EpisodicDataset` classes (imported in place through oracle/refshim.py, with this package's
pure-Python COCO reader standing in for pycocotools and the same deterministic transform callable) when the reference is
present; the hand-computed checks run everywhere."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import refshim

import cape_amd  # noqa: F401
from cape_amd.datasets import (EpisodicDataset, ImageNotFoundError, MP100CAPE, build_episodic_dataloader, episodic_collate_fn)
from cape_amd.datasets.coco_lite import COCO
from cape_amd.datasets.transforms import HostTransform, apply_plan_host, images_from_raw_host, resize_plan, train_plan

needs_ref = pytest.mark.skipif(not refshim.reference_available(), reason="reference checkout not present")

CATS = {1: 5, 2: 9, 3: 4}


def make_dataset(root, n_per_cat=5, seed=0, scale=1):
    """`scale` multiplies the image sizes (bench.py's loader benchmark uses MP-100-like sizes: scale 4 = 240-560 x 200-480)."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(root / "data", exist_ok=True)
    images, anns, aid = [], [], 1
    img_id = 100
    for cat, nk in CATS.items():
        for j in range(n_per_cat):
            w, h = int(rng.integers(60, 140)) * scale, int(rng.integers(50, 120)) * scale
            arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            name = f"c{cat}/img_{img_id}.png"
            os.makedirs(root / "data" / f"c{cat}", exist_ok=True)
            Image.fromarray(arr).save(root / "data" / name)
            images.append({"id": img_id, "file_name": name, "width": w, "height": h})
            bx, by = float(rng.uniform(0, w * 0.2)), float(rng.uniform(0, h * 0.2))
            bw, bh = float(rng.uniform(w * 0.5, w * 0.9)), float(rng.uniform(h * 0.5, h * 0.9))      # may stick out: clamped
            k = []
            for i in range(nk):
                v = int(rng.choice([0, 1, 2], p=[0.2, 0.2, 0.6]))
                k += [float(rng.uniform(bx, bx + bw * 0.95)), float(rng.uniform(by, by + bh * 0.95)), v]
            if not any(k[2::3]):
                k[2] = 2
            anns.append({"id": aid, "image_id": img_id, "category_id": cat, "bbox": [bx, by, bw, bh], "keypoints": k, "num_keypoints": nk})
            aid += 1
            if j == 0:                      # a second instance on the first image of each category (ignored: first one wins)
                anns.append({"id": aid, "image_id": img_id, "category_id": cat, "bbox": [1.0, 1.0, 20.0, 20.0],
                             "keypoints": [5.0, 5.0, 2] * nk, "num_keypoints": nk})
                aid += 1
            img_id += 1
    # a record without visible keypoints and one whose file is missing
    images.append({"id": 900, "file_name": "c1/img_100.png", "width": 10, "height": 10})
    anns.append({"id": aid, "image_id": 900, "category_id": 1, "bbox": [0, 0, 5, 5], "keypoints": [1.0, 1.0, 0] * 5}); aid += 1
    images.append({"id": 901, "file_name": "c1/missing.png", "width": 10, "height": 10})
    anns.append({"id": aid, "image_id": 901, "category_id": 1, "bbox": [0, 0, 5, 5], "keypoints": [1.0, 1.0, 2] * 5}); aid += 1
    cats = [{"id": c, "name": f"cat{c}", "keypoints": [f"k{i}" for i in range(nk)], "skeleton": [[i, i + 1] for i in range(nk - 1)]}
            for c, nk in CATS.items()]
    os.makedirs(root / "annotations", exist_ok=True)
    with open(root / "annotations" / "mp100_split1_train.json", "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": cats}, f)
    with open(root / "category_splits.json", "w") as f:
        json.dump({"train": [1, 2], "val": [3], "test": [3]}, f)
    return root / "annotations" / "mp100_split1_train.json"


def test_record_fields_crop_and_keypoints(tmp_path):
    ann = make_dataset(tmp_path)
    ds = MP100CAPE(str(tmp_path / "data"), str(ann), HostTransform(train=False, size=64), vocab_size=2000, seq_len=200)
    assert len(ds) == 17 and ds.multi_instance_stats["multi_instance_images"] == 3
    coco = COCO(str(ann))
    r = ds[0]
    a = coco.loadAnns(coco.getAnnIds(imgIds=ds.ids[0]))[0]
    info = coco.loadImgs(ds.ids[0])[0]
    bx, by = int(a["bbox"][0]), int(a["bbox"][1])
    bw, bh = min(int(a["bbox"][2]), info["width"] - bx), min(int(a["bbox"][3]), info["height"] - by)
    assert r["bbox"] == [bx, by, bw, bh] and r["bbox_width"] == bw and r["bbox_height"] == bh
    assert r["image"].shape == (3, 64, 64) and r["height"] == 64 and r["width"] == 64 and 0 <= float(r["image"].min()) and float(r["image"].max()) <= 1
    k = np.array(a["keypoints"]).reshape(-1, 3)
    want = np.c_[(k[:, 0] - bx) * 64 / bw, (k[:, 1] - by) * 64 / bh]
    assert np.allclose(np.array(r["keypoints"]), want, atol=1e-9)
    assert r["visibility"] == k[:, 2].tolist() and r["num_keypoints"] == 5 and r["category_id"] == 1
    assert r["skeleton"] == [[i, i + 1] for i in range(4)]
    t = r["seq_data"]
    assert t["seq11"].shape == (200,) and int(t["token_labels"][5]) == 2 and int((t["token_labels"] == 0).sum()) == 5
    assert [bool(b) for b in t["visibility_mask"][:6]] == [v > 0 for v in r["visibility"]] + [True]
    with pytest.raises(ImageNotFoundError):
        ds[ds.ids.index(900)]
    with pytest.raises(ImageNotFoundError):
        ds[ds.ids.index(901)]


def test_transform_plans_move_pixels_and_keypoints_together():
    """A bright blob painted at a keypoint must land where the plan maps the keypoint (random train plans incl. flips)."""
    rng = np.random.default_rng(3)
    for trial in range(6):
        h, w = 90, 130
        img = np.zeros((h, w, 3), dtype=np.uint8)
        kx, ky = 40.3 + trial * 7, 30.7 + trial * 4
        yy, xx = np.mgrid[0:h, 0:w]
        blob = np.exp(-(((xx + 0.5 - kx) ** 2 + (yy + 0.5 - ky) ** 2) / 8.0))
        img[..., 0] = (blob * 255).astype(np.uint8)
        plan = train_plan(h, w, rng, size=128) if trial else resize_plan(h, w, 128)
        plan.color, plan.mode = None, 0                     # geometry only
        out = apply_plan_host(img, plan)[0]
        (mx, my), = plan.map_keypoints([(kx, ky)])
        if not (4 < mx < 124 and 4 < my < 124):
            continue
        m = out / out.sum()
        vy, vx = torch.meshgrid(torch.arange(128.0) + 0.5, torch.arange(128.0) + 0.5, indexing="ij")
        cx, cy = float((m * vx).sum()), float((m * vy).sum())
        assert abs(cx - mx) < 1.0 and abs(cy - my) < 1.0, (trial, cx, cy, mx, my)
    # every branch of the training distribution is drawn and runs: colour orders, noise, Gaussian and motion blur, flips
    crop = rng.integers(0, 256, (70, 50, 3), dtype=np.uint8)
    seen = set()
    g = np.random.default_rng(9)
    for _ in range(60):
        plan = train_plan(70, 50, g, size=64)
        a = apply_plan_host(crop, plan)
        assert a.shape == (3, 64, 64) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-6 and torch.isfinite(a).all()
        seen.add((plan.mode, plan.color is not None, plan.flipped, None if plan.blur_kernel is None else plan.blur_kernel.shape[0]))
        if plan.blur_kernel is not None:
            assert abs(float(plan.blur_kernel.sum()) - 1.0) < 1e-5
    assert {m for m, *_ in seen} == {0, 1, 2} and {k for *_, k in seen} >= {None, 3, 5}
    # identity plan = plain bilinear resize of the crop
    ident = apply_plan_host(crop, resize_plan(70, 50, 64))
    ref = torch.nn.functional.interpolate(torch.from_numpy(crop).permute(2, 0, 1)[None].float() / 255.0, size=(64, 64), mode="bilinear",
                                          align_corners=False)[0]
    assert torch.allclose(ident, ref, atol=1e-6)


def test_episodic_dataset_and_loader(tmp_path):
    ann = make_dataset(tmp_path)
    ds = MP100CAPE(str(tmp_path / "data"), str(ann), HostTransform(train=False, size=64), vocab_size=2000, seq_len=200)
    ep = EpisodicDataset(ds, str(tmp_path / "category_splits.json"), split="train", num_queries_per_episode=2,
                         num_support_per_episode=2, episodes_per_epoch=6, seed=5, fixed_episodes=True)
    assert sorted(ep.sampler.categories) == [1, 2] and len(ep) == 6
    e = ep[0]
    nk = CATS[e["category_id"]]
    assert len(e["support_coords"]) == 2 and e["support_coords"][0].shape == (nk, 2) and e["support_masks"][0].dtype == torch.bool
    assert float(e["support_coords"][0].min()) >= 0 and float(e["support_coords"][0].max()) <= 1
    assert len(e["query_images"]) == 2 and e["query_images"][0].shape == (3, 64, 64) and len(e["query_metadata"][0]["visibility"]) == nk
    assert e["support_images"][0].shape == (3, 64, 64)
    b = episodic_collate_fn([ep[0], ep[1], ep[2]])
    P = max(CATS[x["category_id"]] for x in (ep[0], ep[1], ep[2]))
    assert b["support_coords"].shape == (6, P, 2) and b["query_images"].shape == (6, 3, 64, 64) and b["category_ids"].shape == (6,)
    # categories with too few examples are dropped / rejected
    with pytest.raises(ValueError):
        EpisodicDataset(ds, str(tmp_path / "category_splits.json"), split="val", num_queries_per_episode=5, num_support_per_episode=5)
    dl = build_episodic_dataloader(ds, str(tmp_path / "category_splits.json"), split="train", batch_size=2, episodes_per_epoch=4,
                                   num_workers=0, seed=1)
    nb = sum(1 for _ in dl)
    assert nb == 2
    # deferred pixels: raw crops + plans travel, the pipeline makes the batch (its host counterpart here; the GPU one in tests/test_augment_gpu.py)
    ds2 = MP100CAPE(str(tmp_path / "data"), str(ann), HostTransform(train=False, size=64), vocab_size=2000, seq_len=200, defer_pixels=True)
    ep2 = EpisodicDataset(ds2, str(tmp_path / "category_splits.json"), split="train", episodes_per_epoch=2, seed=5, fixed_episodes=True,
                          load_support_images=False)
    ep1 = EpisodicDataset(ds, str(tmp_path / "category_splits.json"), split="train", episodes_per_epoch=2, seed=5, fixed_episodes=True)
    b2, b1 = episodic_collate_fn([ep2[0]]), episodic_collate_fn([ep1[0]])
    assert b2["query_images"] is None and len(b2["query_raw"]) == 2
    imgs = images_from_raw_host([c for c, _ in b2["query_raw"]], [p for _, p in b2["query_raw"]])
    assert torch.allclose(imgs, b1["query_images"], atol=1e-6)
    for k in b1["query_targets"]:
        assert torch.equal(b1["query_targets"][k], b2["query_targets"][k]), k


def _ref_classes():
    refshim.install()
    import importlib
    m = importlib.import_module("datasets.mp100_cape")
    m.COCO = COCO                                   # pycocotools stand-in: this package's pure-Python reader
    s = importlib.import_module("datasets.episodic_sampler")
    return m, s


@needs_ref
def test_records_and_episodes_match_reference_classes(tmp_path):
    ann = make_dataset(tmp_path)
    m, s = _ref_classes()
    tr = HostTransform(train=False, size=64)
    ref_ds = m.MP100CAPE(str(tmp_path / "data"), str(ann), tr, vocab_size=2000, seq_len=200)
    ds = MP100CAPE(str(tmp_path / "data"), str(ann), tr, vocab_size=2000, seq_len=200)
    assert ref_ds.ids == ds.ids and ref_ds.multi_instance_stats == ds.multi_instance_stats
    for i in range(len(ds)):
        try:
            want = ref_ds[i]
        except m.ImageNotFoundError:
            with pytest.raises(ImageNotFoundError):
                ds[i]
            continue
        got = ds[i]
        for k in ("image_id", "category_id", "num_keypoints", "num_visible_keypoints", "bbox", "bbox_width", "bbox_height", "height",
                  "width", "visibility", "skeleton"):
            assert got[k] == want[k], (i, k)
        assert np.allclose(np.array(got["keypoints"]), np.array(want["keypoints"]), atol=1e-9)
        assert torch.equal(got["image"], want["image"])
        for k in want["seq_data"]:
            assert torch.equal(got["seq_data"][k], want["seq_data"][k].to(got["seq_data"][k].dtype)), (i, k)
    # same seeds -> same episodes (python `random` drives the sampler in both)
    split = str(tmp_path / "category_splits.json")
    ref_ep = s.EpisodicDataset(ref_ds, split, split="train", num_queries_per_episode=2, num_support_per_episode=2, episodes_per_epoch=5, seed=11, fixed_episodes=True)
    ep = EpisodicDataset(ds, split, split="train", num_queries_per_episode=2, num_support_per_episode=2, episodes_per_epoch=5, seed=11, fixed_episodes=True)
    assert ref_ep._cached_episodes == ep._cached_episodes
    for i in range(5):
        random.seed(100 + i)                    # a fixed episode with an unusable image falls back to random sampling
        a = ref_ep[i]
        random.seed(100 + i)
        b = ep[i]
        assert a["category_id"] == b["category_id"] and a["support_skeletons"] == b["support_skeletons"]
        for x, y in zip(a["support_coords"], b["support_coords"]):
            assert torch.equal(x, y)
        for x, y in zip(a["support_masks"], b["support_masks"]):
            assert torch.equal(x, y)
        for x, y in zip(a["query_images"], b["query_images"]):
            assert torch.equal(x, y)
        for x, y in zip(a["query_metadata"], b["query_metadata"]):
            for k in ("image_id", "height", "width", "num_keypoints", "num_visible_keypoints", "bbox", "bbox_width", "bbox_height", "visibility"):
                assert x[k] == y[k], k
        assert a["support_metadata"] == b["support_metadata"]


class _PlanProbe(torch.utils.data.Dataset):
    """Each item reports which worker made it and the first numbers of the plan it drew."""

    def __init__(self, tr):
        self.tr = tr

    def __len__(self):
        return 8

    def __getitem__(self, i):
        info = torch.utils.data.get_worker_info()
        p = self.tr.plan(60, 80)
        return torch.tensor([-1 if info is None else info.id, i], dtype=torch.float64), torch.from_numpy(np.r_[p.fwd.reshape(-1), float(p.noise_seed)])


def test_augmentation_streams_differ_per_worker_rank_and_epoch():
    """the plan generator is seeded in the process that draws from it, from (seed, rank, torch's per-worker,
    per-epoch seed).  One generator created in the parent and inherited by every forked worker repeated the same stream in every
    worker, on every rank, in every epoch."""
    tr = HostTransform(train=True, size=64, seed=7, rank=0)
    dl = torch.utils.data.DataLoader(_PlanProbe(tr), batch_size=1, num_workers=2, shuffle=False)

    def epoch():
        first = {}
        for who, plan in dl:
            first.setdefault(int(who[0, 0]), plan[0].clone())
        return first

    e1, e2 = epoch(), epoch()
    assert set(e1) == {0, 1}
    assert not torch.equal(e1[0], e1[1]), "two workers drew the same first plan"
    assert not torch.equal(e1[0], e2[0]) and not torch.equal(e1[1], e2[1]), "epoch 2 repeated epoch 1's plans"
    # ranks differ for the same seed and worker seed; the same (seed, rank) in one process is reproducible
    a, b = HostTransform(train=True, size=64, seed=7, rank=0), HostTransform(train=True, size=64, seed=7, rank=1)
    a2 = HostTransform(train=True, size=64, seed=7, rank=0)
    pa, pb, pa2 = a.plan(60, 80), b.plan(60, 80), a2.plan(60, 80)
    assert pa.noise_seed == pa2.noise_seed and np.array_equal(pa.fwd, pa2.fwd)
    assert pa.noise_seed != pb.noise_seed
