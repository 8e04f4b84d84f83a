"""`evaluate_cape` glue pinned to the reference (reference `models/engine_cape.py:394-870`): the
fixture tests/golden/eval_glue.{npz,json} holds crafted autoregressive predictions -- early <eos> (zero padding), excess
keypoints (trim to the category's count), a <sep> inside the stream, ragged categories in one batch, T < L and T = L, a
batch without query_metadata -- and what the REAL reference's `evaluate_cape` returned for them
(oracle/make_golden_r2.py).  CPU only: the PCK side of the function is host logic; the validation loss goes through the
HIP criterion and is checked in tests/test_e2e_gpu.py."""
import json
import os

import numpy as np
import torch

import cape_amd  # noqa: F401
from cape_amd.models.engine_cape import evaluate_cape, extract_keypoints_from_predictions, extract_keypoints_from_sequence


def load_eval_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "eval_glue.npz"))
    meta = json.load(open(os.path.join(golden_dir, "eval_glue.json")))
    batches, preds = [], []
    for i in range(meta["n_batches"]):
        tg = {k[len(f"b{i}_t_"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith(f"b{i}_t_")}
        b = {"support_coords": torch.from_numpy(d[f"b{i}_support_coords"]), "support_masks": torch.from_numpy(d[f"b{i}_support_masks"]),
             "query_images": torch.zeros(tg["seq11"].shape[0], 3, 8, 8), "support_skeletons": None, "query_targets": tg,
             "category_ids": torch.from_numpy(d[f"b{i}_category_ids"])}
        if meta["query_metadata"][i] is not None:
            b["query_metadata"] = meta["query_metadata"][i]
        batches.append(b)
        lg = torch.from_numpy(d[f"b{i}_logits"])
        preds.append({"logits": lg, "coordinates": torch.from_numpy(d[f"b{i}_coordinates"]), "sequences": lg.argmax(-1)})
    return batches, preds, meta


class FakeModel(torch.nn.Module):
    def __init__(self, preds, device="cpu"):
        super().__init__()
        self.preds, self.calls, self.device = preds, 0, device

    def forward_inference(self, samples, support_coords, support_mask, skeleton_edges=None):
        p = {k: v.to(self.device) for k, v in self.preds[self.calls].items()}
        self.calls += 1
        return p


def test_evaluate_cape_pck_glue_matches_reference(golden_dir):
    batches, preds, meta = load_eval_fixture(golden_dir)
    stats = evaluate_cape(FakeModel(preds), None, batches, torch.device("cpu"), compute_pck=True, pck_threshold=0.2)
    ref = meta["stats_no_criterion"]
    assert stats["pck_num_correct"] == ref["pck_num_correct"] and stats["pck_num_visible"] == ref["pck_num_visible"]
    assert abs(stats["pck"] - ref["pck"]) < 1e-12 and abs(stats["pck_mean_categories"] - ref["pck_mean_categories"]) < 1e-12
    assert stats["loss"] == 0.0 and stats["loss_ce"] == 0.0 and stats["loss_coords"] == 0.0       # no criterion: zeros, as the reference
    for i, want in enumerate(meta["per_batch_correct_visible"]):
        m = FakeModel(preds); m.calls = i
        s = evaluate_cape(m, None, batches[i:i + 1], torch.device("cpu"))
        assert [s["pck_num_correct"], s["pck_num_visible"]] == want, (i, s, want)


def test_keypoint_extraction_rules(golden_dir):
    """GT keypoints by GT labels under the mask, predicted keypoints by argmax == <coord> (a <sep> is skipped, everything
    after <eos> that is typed <coord> still counts: the reference does not stop at <eos>, util/sequence_utils.py:8-65)."""
    batches, preds, _ = load_eval_fixture(golden_dir)
    t = batches[0]["query_targets"]
    gt = extract_keypoints_from_sequence(t["target_seq"], t["token_labels"], t["mask"])
    n_gt = [(t["token_labels"][i][t["mask"][i]] == 0).sum().item() for i in range(6)]
    assert n_gt == [5, 5, 9, 9, 17, 17] and gt.shape == (6, 17, 2) and float(gt[0, 5:].abs().sum()) == 0
    pk = extract_keypoints_from_predictions(preds[0]["coordinates"], preds[0]["logits"])
    n_pred = [(preds[0]["logits"][i].argmax(-1) == 0).sum().item() for i in range(6)]
    assert n_pred == [5, 3, 9, 14, 17, 10] and pk.shape == (6, 17, 2)
    assert torch.equal(pk[2, :9], preds[0]["coordinates"][2][preds[0]["logits"][2].argmax(-1) == 0])
