"""MP-100 dataset for category-agnostic pose estimation: host data path of the CAPE pipeline (reference
`datasets/mp100_cape.py:71-492`, `build_mp100_cape` :835-962).  One record per image: crop to the first annotated
instance's bbox, keypoints relative to the crop (ALL keypoints kept, visibility carried along), transform to the network
resolution, tokenise (`keypoint_tokenization.tokenize_keypoints` = `_tokenize_keypoints` :625-832).

Differences in *how* (records identical for a given transform): annotations are read by `coco_lite.COCO` (pure Python,
no pycocotools); transforms are plans (`transforms.py`) that run on the host or on the GPU; with `defer_pixels=True` the
record carries the raw uint8 crop + plan instead of the finished image, and `transforms.DeviceImagePipeline` produces the
pixels on the device."""
import os
from pathlib import Path

import numpy as np
import torch

from .coco_lite import COCO
from .discrete_tokenizer import DiscreteTokenizerV2
from .keypoint_tokenization import tokenize_keypoints
from .transforms import HostTransform


class ImageNotFoundError(Exception):
    """Raised for a record that cannot be used (missing / empty image, no valid annotation, empty crop): the episodic
    dataset resamples (`episodic_sampler.py:358-364`)."""


class MP100CAPE(torch.utils.data.Dataset):
    def __init__(self, img_folder, ann_file, transforms, semantic_classes=-1, dataset_name="mp100", image_norm=False,
                 poly2seq=True, converter_version="v3", split="train", defer_pixels=False, **kwargs):
        super().__init__()
        self.root = img_folder
        self._transforms = transforms
        self.semantic_classes = semantic_classes
        self.dataset_name = dataset_name
        self.split = split
        self.coco = COCO(ann_file)
        self.ids = list(sorted(self.coco.imgs.keys()))
        self.poly2seq = poly2seq
        self.defer_pixels = defer_pixels
        self.image_norm = image_norm
        self._mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        self._std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        if poly2seq:
            num_bins = int(np.sqrt(kwargs.get("vocab_size", 2000)))
            self.tokenizer = DiscreteTokenizerV2(num_bins=num_bins, seq_len=kwargs.get("seq_len", 200), add_cls=False)
        self._analyze_multi_instance_stats()

    def _analyze_multi_instance_stats(self):
        total_instances = multi = max_inst = 0
        for img_id in self.ids:
            valid = 0
            for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)):
                if "keypoints" in ann and ann["keypoints"] and "bbox" in ann:
                    if np.any(np.array(ann["keypoints"]).reshape(-1, 3)[:, 2] > 0):
                        valid += 1
            if valid > 0:
                total_instances += valid
                multi += valid > 1
                max_inst = max(max_inst, valid)
        n = len(self.ids)
        self.multi_instance_stats = {"total_images": n, "total_instances": total_instances, "multi_instance_images": int(multi),
                                     "max_instances_per_image": max_inst, "instances_used": n,
                                     "instances_unused": total_instances - n}

    def get_vocab_size(self):
        return len(self.tokenizer) if self.poly2seq else None

    def get_tokenizer(self):
        return self.tokenizer if self.poly2seq else None

    def __len__(self):
        return len(self.ids)

    def _get_skeleton_for_category(self, category_id):
        try:
            return self.coco.loadCats(category_id)[0].get("skeleton", []) or []
        except Exception:
            return []

    def _get_num_keypoints_for_category(self, category_id):
        try:
            names = self.coco.loadCats(category_id)[0].get("keypoints", [])
            return len(names) if names else None
        except Exception:
            return None

    def _tokenize_keypoints(self, keypoints, height, width, visibility=None):
        return tokenize_keypoints(self.tokenizer, keypoints, height, width, visibility,
                                  category_id=getattr(self, "_current_category_id", 0))

    def __getitem__(self, index):
        from PIL import Image
        coco = self.coco
        img_id = self.ids[index]
        target = coco.loadAnns(coco.getAnnIds(imgIds=img_id))
        file_name = os.path.join(self.root, coco.loadImgs(img_id)[0]["file_name"])
        if not os.path.exists(file_name):
            raise ImageNotFoundError(f"Image not found: {file_name}")
        img = np.array(Image.open(file_name).convert("RGB"))
        if img is None or img.size == 0 or img.ndim < 2:
            raise ImageNotFoundError(f"Image {img_id} ({file_name}) failed to load or is empty")
        orig_h, orig_w = img.shape[:2]
        record = {"file_name": file_name, "image_id": img_id}
        inst = None
        for ann in target:                                   # first instance with a bbox and >= 1 visible keypoint
            if "keypoints" in ann and ann["keypoints"]:
                kpts = np.array(ann["keypoints"]).reshape(-1, 3)
                if (kpts[:, 2] > 0).any() and "bbox" in ann:
                    inst = (kpts, ann)
                    break
        if inst is None:
            raise ImageNotFoundError(f"Image {img_id} has no valid annotations (no visible keypoints or missing bbox).")
        kpts, ann = inst
        bx, by, bw, bh = ann["bbox"]
        bx, by = max(0, int(bx)), max(0, int(by))
        bw, bh = min(int(bw), orig_w - bx), min(int(bh), orig_h - by)
        crop = img[by:by + bh, bx:bx + bw]
        if crop.size == 0 or crop.shape[0] == 0 or crop.shape[1] == 0:
            raise ImageNotFoundError(f"Image {img_id} produced empty crop with bbox [{bx}, {by}, {bw}, {bh}]. "
                                     f"Original image size: {orig_w}x{orig_h}")
        k = np.array(kpts[:, :2].tolist())                   # ALL keypoints (the reference keeps invisible ones too)
        k[:, 0] -= bx
        k[:, 1] -= by
        vis = kpts[:, 2]
        cat = ann.get("category_id", 0)
        record.update(keypoints=k.tolist(), visibility=vis.tolist(), category_id=cat, num_keypoints=len(k),
                      num_visible_keypoints=int(np.sum(vis > 0)), bbox=[bx, by, bw, bh], bbox_width=bw, bbox_height=bh,
                      height=bh, width=bw, skeleton=self._get_skeleton_for_category(cat))
        img = crop
        if self._transforms is not None:
            try:
                n_before = len(record["keypoints"])
                if self.defer_pixels and hasattr(self._transforms, "plan"):
                    plan = self._transforms.plan(bh, bw)
                    record["keypoints"] = plan.map_keypoints(record["keypoints"])
                    record["raw_crop"], record["plan"] = torch.from_numpy(np.ascontiguousarray(crop)), plan
                    record["height"] = record["width"] = plan.out_size
                    img = None
                else:
                    tr = self._transforms(image=img, keypoints=record["keypoints"])
                    img = tr["image"]
                    kp = tr.get("keypoints", record["keypoints"])
                    if len(kp) != n_before:
                        raise ImageNotFoundError(f"transform dropped keypoints ({n_before} -> {len(kp)})")
                    record["keypoints"] = list(kp)
                    record["height"], record["width"] = img.shape[:2]
            except ImageNotFoundError:
                raise
            except Exception as e:
                raise ImageNotFoundError(f"Image {img_id} ({file_name}) failed during transforms: {e}") from e
        if img is not None:
            t = torch.as_tensor(img[None] if img.ndim == 2 else np.ascontiguousarray(img.transpose(2, 0, 1))).float() / 255.0
            record["image"] = (t - self._mean) / self._std if self.image_norm else t
        else:
            record["image"] = None
        if self.poly2seq:
            self._current_category_id = record["category_id"]
            record["seq_data"] = self._tokenize_keypoints(record["keypoints"], record["height"], record["width"],
                                                          record.get("visibility"))
            del self._current_category_id
        nk, nv = len(record["keypoints"]), len(record["visibility"])
        if nk != nv:
            raise ValueError(f"keypoints length ({nk}) != visibility length ({nv}) for image {img_id}, category {cat}")
        exp = self._get_num_keypoints_for_category(cat)
        if exp is not None and nk != exp:
            raise ValueError(f"keypoints length ({nk}) != expected for category {cat} ({exp}) for image {img_id}")
        return record


def build_mp100_cape(image_set, args, defer_pixels=False):
    """`build_mp100_cape` of the reference (:835-962): annotation file searched under data/cleaned_annotations,
    clean_annotations, annotations of `args.dataset_root`; train = random plan (affine / flip / jitter / noise) + resize to
    512, val / test = resize to 512."""
    split_num = getattr(args, "mp100_split", 1)
    root = Path(args.dataset_root).resolve()
    cands = [root / "data" / "cleaned_annotations" / f"mp100_split{split_num}_{image_set}.json",
             root / "clean_annotations" / f"mp100_split{split_num}_{image_set}.json",
             root / "annotations" / f"mp100_split{split_num}_{image_set}.json"]
    ann = next((p for p in cands if p.exists()), None)
    if ann is None:
        raise FileNotFoundError("Annotation file not found in any location:\n" + "\n".join(f"  - {p}" for p in cands))
    from ..util import misc as utils
    tr = HostTransform(train=(image_set == "train"), size=512, seed=getattr(args, "seed", None), rank=utils.get_rank())
    return MP100CAPE(img_folder=str(Path(args.dataset_root) / "data"), ann_file=str(ann), transforms=tr,
                     semantic_classes=args.semantic_classes, dataset_name="mp100", image_norm=args.image_norm, poly2seq=True,
                     converter_version="v3", split=image_set, vocab_size=args.vocab_size, seq_len=args.seq_len,
                     defer_pixels=defer_pixels)
