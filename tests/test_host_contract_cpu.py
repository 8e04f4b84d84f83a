"""Host-side contract of the path: the episodic collate (K-shot mean-pooling, padding,
repeat per query) and the PCK evaluator, cross-checked against the reference's own functions when the reference is
importable (the build container); on a machine without /root/reference the cross-checks skip and the golden-vector tests
(tests/test_oracle_golden.py: tokenizer.npz, pck.npz) remain."""
import numpy as np
import pytest
import torch

from oracle import refshim

import cape_amd  # noqa: F401
from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn
from cape_amd.datasets.synthetic import SyntheticEpisodes
from cape_amd.util.eval_utils import PCKEvaluator, compute_pck_bbox

needs_ref = pytest.mark.skipif(not refshim.reference_available(), reason="reference checkout not present")


def _same(a, b, path=""):
    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), path
    elif isinstance(a, dict):
        assert set(a) == set(b), path
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, np.ndarray):
        assert np.array_equal(a, b), path
    else:
        assert a == b, path


@needs_ref
@pytest.mark.parametrize("n_support", [1, 5])
def test_collate_matches_reference(n_support):
    """datasets/episodic_sampler.py:372-482: ragged keypoint counts across episodes, K-shot supports."""
    refshim.install()
    from datasets.episodic_sampler import episodic_collate_fn as ref_collate
    tok = DiscreteTokenizerV2(44, 200)
    a = SyntheticEpisodes(tok, 4, 64, 9, 2, num_support=n_support, seed=4)
    b = SyntheticEpisodes(tok, 4, 64, 5, 2, num_support=n_support, seed=5)          # fewer keypoints: padding path
    items = [a[0], b[1], a[2], b[3]]
    _same(ref_collate(items), episodic_collate_fn(items))


@needs_ref
def test_pck_evaluator_matches_reference():
    """util/eval_utils.py:29-260 on random predictions: tensors and lists, visibility 0/1/2, three categories."""
    refshim.install()
    from util.eval_utils import PCKEvaluator as RefEval
    rng = np.random.Generator(np.random.PCG64(9))
    mine, ref = PCKEvaluator(0.2), RefEval(0.2)
    for it in range(3):
        B, P = 5, 11
        gt = torch.from_numpy(rng.random((B, P, 2)).astype(np.float32) * 200)
        pred = gt + torch.from_numpy(rng.normal(0, 25, (B, P, 2)).astype(np.float32))
        vis = torch.from_numpy(rng.integers(0, 3, (B, P)))
        bw, bh = torch.from_numpy(rng.uniform(64, 300, B).astype(np.float32)), torch.from_numpy(rng.uniform(64, 300, B).astype(np.float32))
        cats = torch.from_numpy(rng.integers(1, 4, B))
        if it == 1:       # list form with ragged lengths
            lens = [P, 7, 3, P, 9]
            args = ([pred[i, :n] for i, n in enumerate(lens)], [gt[i, :n] for i, n in enumerate(lens)], bw, bh, cats,
                    [vis[i, :n] for i, n in enumerate(lens)])
        else:
            args = (pred, gt, bw, bh, cats, vis)
        mine.add_batch(*args)
        ref.add_batch(*args)
    rm, rr = mine.get_results(), ref.get_results()
    assert rm["total_correct"] == rr["total_correct"] and rm["total_visible"] == rr["total_visible"]
    assert rm["pck_per_category"].keys() == rr["pck_per_category"].keys()
    for k in ("pck_overall", "mean_pck_categories"):
        assert abs(rm[k] - rr[k]) < 1e-12
    assert [r["num_correct"] for r in mine.image_results] == [r["num_correct"] for r in ref.image_results]


def test_pck_edge_cases():
    """No visible keypoint -> (0.0, 0, 0); the threshold comparison is strict (util/eval_utils.py: `normalized < threshold`)."""
    gt = np.zeros((3, 2), dtype=np.float32)
    pck, c, v = compute_pck_bbox(gt + 1.0, gt, 10.0, 10.0, visibility=np.zeros(3), threshold=0.2)
    assert (pck, c, v) == (0.0, 0, 0)
    pred = np.array([[9.999, 0.0], [10.0, 0.0], [10.001, 0.0]], dtype=np.float64)          # bbox diagonal 50 -> 10 px at 0.2
    pck, c, v = compute_pck_bbox(pred, np.zeros((3, 2)), 30.0, 40.0, visibility=np.array([2, 1, 2]), threshold=0.2)
    assert (c, v) == (1, 3) and abs(pck - 1.0 / 3.0) < 1e-12
