"""PCK@bbox metric (interface of the reference's `util/eval_utils.py:29-110, :158-269`): a keypoint is
correct when ||pred - gt|| / bbox_size < threshold, counted over keypoints with visibility > 0; overall
(micro) PCK and per-category (macro) mean.  Host NumPy float64 -- the metric is not a kernel.
`PCKEvaluator.synchronize_between_processes` (new) sums the counters over ranks for data-parallel eval."""
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def compute_pck_bbox(pred_keypoints, gt_keypoints, bbox_width, bbox_height, visibility=None, threshold=0.2,
                     normalize_by="diagonal"):
    pred, gt = _np(pred_keypoints), _np(gt_keypoints)
    assert pred.shape == gt.shape, f"Shape mismatch: pred {pred.shape} vs gt {gt.shape}"
    assert pred.ndim == 2 and pred.shape[1] == 2, f"Expected (N, 2) keypoints, got {pred.shape}"
    n = len(pred)
    if visibility is None:
        vis = np.ones(n, dtype=bool)
    else:
        v = np.array(_np(visibility))
        assert len(v) == n, f"Visibility length ({len(v)}) must match keypoints ({n})"
        vis = v > 0
    num_visible = int(vis.sum())
    if num_visible == 0:
        return 0.0, 0, 0
    p, g = pred[vis], gt[vis]
    if np.allclose(p, g, atol=1e-6):
        warnings.warn("Predictions are IDENTICAL to ground truth! This indicates data leakage or teacher forcing "
                      "during evaluation.", RuntimeWarning)
    dist = np.sqrt(np.sum((p - g) ** 2, axis=1))
    if normalize_by == "diagonal":
        size = np.sqrt(bbox_width ** 2 + bbox_height ** 2)
    elif normalize_by == "max":
        size = max(bbox_width, bbox_height)
    elif normalize_by == "mean":
        size = (bbox_width + bbox_height) / 2
    else:
        raise ValueError(f"Unknown normalize_by: {normalize_by}")
    num_correct = int((dist / size < threshold).sum())
    return float(num_correct / num_visible), num_correct, num_visible


class PCKEvaluator:
    def __init__(self, threshold: float = 0.2, normalize_by: str = "diagonal"):
        self.threshold, self.normalize_by = threshold, normalize_by
        self.reset()

    def reset(self):
        self.total_correct = 0
        self.total_visible = 0
        self.category_correct = {}
        self.category_visible = {}
        self.image_results = []
        self.num_images_global = None      # set by synchronize_between_processes (image_results stays process-local)

    def add_batch(self, pred_keypoints, gt_keypoints, bbox_widths, bbox_heights, category_ids=None, visibility=None,
                  image_ids: Optional[List] = None):
        n = len(pred_keypoints)
        for i in range(n):
            vis = visibility[i] if visibility is not None else None
            cat = int(category_ids[i]) if category_ids is not None else 0
            pck, correct, visible = compute_pck_bbox(pred_keypoints[i], gt_keypoints[i], float(bbox_widths[i]),
                                                     float(bbox_heights[i]), visibility=vis, threshold=self.threshold,
                                                     normalize_by=self.normalize_by)
            self.total_correct += correct
            self.total_visible += visible
            self.category_correct[cat] = self.category_correct.get(cat, 0) + correct
            self.category_visible[cat] = self.category_visible.get(cat, 0) + visible
            self.image_results.append({"image_id": image_ids[i] if image_ids is not None else None, "category_id": cat,
                                       "pck": pck, "num_correct": correct, "num_visible": visible})

    def synchronize_between_processes(self, max_categories=128):
        """Sum the counters over ranks with one small all-reduce (the reference's evaluator is process-local)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        buf = torch.zeros(3 + 2 * max_categories, dtype=torch.float64, device=dev)
        buf[0], buf[1] = self.total_correct, self.total_visible
        buf[2 + 2 * max_categories] = len(self.image_results)
        for c in self.category_correct:
            assert 0 <= c < max_categories
            buf[2 + c] = self.category_correct[c]
            buf[2 + max_categories + c] = self.category_visible[c]
        dist.all_reduce(buf)
        b = buf.cpu().tolist()
        self.total_correct, self.total_visible = int(b[0]), int(b[1])
        self.num_images_global = int(b[2 + 2 * max_categories])
        self.category_correct = {c: int(b[2 + c]) for c in range(max_categories) if b[2 + max_categories + c] > 0 or b[2 + c] > 0}
        self.category_visible = {c: int(b[2 + max_categories + c]) for c in self.category_correct}

    def get_results(self) -> Dict:
        per = {c: (self.category_correct[c] / self.category_visible[c] if self.category_visible[c] > 0 else 0.0)
               for c in self.category_correct}
        return {"pck_overall": self.total_correct / self.total_visible if self.total_visible > 0 else 0.0,
                "pck_per_category": per, "mean_pck_categories": float(np.mean(list(per.values()))) if per else 0.0,
                "total_correct": self.total_correct, "total_visible": self.total_visible, "num_categories": len(per),
                "num_images": self.num_images_global if self.num_images_global is not None else len(self.image_results), "threshold": self.threshold}
