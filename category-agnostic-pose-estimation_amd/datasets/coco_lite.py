"""Minimal reader of COCO-style keypoint annotation files: the subset of `pycocotools.coco.COCO` that the MP-100 loader
touches (`datasets/mp100_cape.py:99-100,213-218,505-524`; `episodic_sampler.py:60-67`): `imgs`, `getAnnIds(imgIds=)`,
`loadAnns`, `loadImgs`, `loadCats`.  Pure Python / json: pycocotools is a compiled third-party package that the path does
not need (no mask decoding, no evaluation)."""
import json
from collections import defaultdict


class COCO:
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns = defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file) as f:
                self.dataset = json.load(f)
            if not isinstance(self.dataset, dict):
                raise TypeError(f"annotation file format {type(self.dataset)} not supported")
            self.createIndex()

    def createIndex(self):
        for ann in self.dataset.get("annotations", []):
            self.imgToAnns[ann["image_id"]].append(ann)
            self.anns[ann["id"]] = ann
        for img in self.dataset.get("images", []):
            self.imgs[img["id"]] = img
        for cat in self.dataset.get("categories", []):
            self.cats[cat["id"]] = cat

    @staticmethod
    def _aslist(x):
        return x if isinstance(x, (list, tuple)) else [x]

    def getAnnIds(self, imgIds=(), catIds=(), iscrowd=None):
        imgIds, catIds = self._aslist(imgIds), self._aslist(catIds)
        if len(imgIds):
            anns = [a for i in imgIds for a in self.imgToAnns.get(i, [])]
        else:
            anns = self.dataset.get("annotations", [])
        if len(catIds):
            anns = [a for a in anns if a.get("category_id") in catIds]
        if iscrowd is not None:
            anns = [a for a in anns if a.get("iscrowd", 0) == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids=()):
        return [self.anns[i] for i in self._aslist(ids)]

    def loadImgs(self, ids=()):
        return [self.imgs[i] for i in self._aslist(ids)]

    def loadCats(self, ids=()):
        return [self.cats[i] for i in self._aslist(ids)]
