"""Row f2 on the GPU: `cape_augment_batch` (csrc/augment.hip: warp + flip, colour jitter, blur / noise, resize, normalise -- two
launches per batch) produces the pixels of the host implementation of the same plans (datasets/transforms.apply_plan_host), for
every branch of the training distribution and ragged crop sizes; and the engine makes its query batch from deferred records."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_augmentation_equals_host_pixels():
    import cape_amd  # noqa: F401
    from cape_amd.datasets.transforms import DeviceImagePipeline, TransformPlan, _gauss_kernel, _motion_kernel, apply_plan_host, train_plan
    rng = np.random.default_rng(5)
    crops, plans = [], []
    for i in range(40):                                      # random draws: all colour orders, noise, Gaussian / motion blur, flips
        h, w = int(rng.integers(9, 200)), int(rng.integers(9, 260))
        crops.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        plans.append(train_plan(h, w, rng, size=96))
    # forced corner cases: a 7 x 7 blur on a crop narrower than the kernel radius allows to reflect once, hue on grey pixels, noise
    crops.append(rng.integers(0, 256, (3, 2, 3), dtype=np.uint8))
    plans.append(TransformPlan(3, 2, 96, mode=2, blur_kernel=_gauss_kernel(7)))
    crops.append(np.full((20, 30, 3), 128, dtype=np.uint8))
    plans.append(TransformPlan(20, 30, 96, color=([3, 1, 0, 2], 1.2, 0.8, 1.3, 0.07)))
    crops.append(rng.integers(0, 256, (64, 48, 3), dtype=np.uint8))
    plans.append(TransformPlan(64, 48, 96, mode=1, noise_std=0.02, noise_seed=12345))
    crops.append(rng.integers(0, 256, (64, 48, 3), dtype=np.uint8))
    plans.append(TransformPlan(64, 48, 96, mode=2, blur_kernel=_motion_kernel(5, rng), color=([0, 1, 2, 3], 0.75, 1.25, 0.7, -0.1)))
    modes = {(p.mode, p.color is not None) for p in plans}
    assert {m for m, _ in modes} == {0, 1, 2} and {c for _, c in modes} == {True, False}
    pipe = DeviceImagePipeline("cuda", out_size=96)
    got = pipe(crops, plans).cpu()
    torch.cuda.synchronize()
    worst = 0.0
    for i, (c, p) in enumerate(zip(crops, plans)):
        want = apply_plan_host(c, p)
        err = float((got[i] - want).abs().max())
        worst = max(worst, err)
        assert err < 2e-4, (i, err, p.mode, p.color)
    # normalisation epilogue, and a second batch through the same pipeline (staging buffers are re-made per batch)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    pn = DeviceImagePipeline("cuda", out_size=96, mean=mean, std=std)
    got2 = pn(crops[:5], plans[:5]).cpu()
    for i in range(5):
        want = (apply_plan_host(crops[i], plans[i]) - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
        assert float((got2[i] - want).abs().max()) < 1e-3
    assert pipe([], []).shape == (0, 3, 96, 96)


def test_engine_builds_the_query_batch_from_deferred_records(tmp_path):
    """The MP-100 path as the CLI runs it: MP100CAPE(defer_pixels=True) -> EpisodicDataset -> collate (raw crops + plans) ->
    engine._to_device -> DeviceImagePipeline; same images as the host-transform dataset for the deterministic validation plans."""
    import cape_amd  # noqa: F401
    from cape_amd.datasets import EpisodicDataset, MP100CAPE, episodic_collate_fn
    from cape_amd.datasets.transforms import HostTransform
    from cape_amd.models.engine_cape import _to_device
    from tests.test_data_path_cpu import make_dataset
    ann = make_dataset(tmp_path)
    mk = lambda defer: EpisodicDataset(MP100CAPE(str(tmp_path / "data"), str(ann), HostTransform(train=False, size=64), vocab_size=2000,
                                                 seq_len=200, defer_pixels=defer),
                                       str(tmp_path / "category_splits.json"), split="train", episodes_per_epoch=2, seed=5,
                                       fixed_episodes=True, load_support_images=False)
    b_dev, b_host = episodic_collate_fn([mk(True)[0], mk(True)[1]]), episodic_collate_fn([mk(False)[0], mk(False)[1]])
    assert b_dev["query_images"] is None and len(b_dev["query_raw"]) == 4
    _, _, imgs, _, tg = _to_device(b_dev, torch.device("cuda"))
    assert imgs.shape == (4, 3, 64, 64) and imgs.is_cuda
    assert float((imgs.cpu() - b_host["query_images"]).abs().max()) < 2e-4
    for k in tg:
        assert torch.equal(tg[k].cpu(), b_host["query_targets"][k])
