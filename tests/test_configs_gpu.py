"""GPU parity at the geometries of BASELINE.json configs[3] and configs[4], and the engine loops pinned to the
reference: product path (HIP kernels behind the C ABI) against fixtures emitted by the REAL
reference (oracle/make_golden_r2.py).  Tolerances of the north star: logits 1e-3, coordinates 1e-4, argmax tokens exact
where the reference's top-2 margin exceeds the logit tolerance."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cape_ref, procweights, synth
from tests.helpers import build_product, cfg5_episode_batch, proc_sd_512, to_dev, train_loop_batches

CFG = cape_ref.Cfg()


@pytest.fixture(params=["bf16x3", "f32"], autouse=True)
def gemm_precision(request):
    from cape_amd.hip import ops
    old = ops.get_gemm_precision()
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision(old)


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_cfg4_384_forward_loss_grads(golden_dir, proc_sd):
    """configs[3] minus Swin-T: ResNet-50 at 384x384, S = 3060 tokens (the MSDA backward runs its 4-channel slab form)."""
    d = np.load(os.path.join(golden_dir, "cfg4_384.npz"))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    model.eval()
    b = to_dev(synth.make_batch(31, 1, 2, 384, 17, CFG, n_invisible=(2,)))
    out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                targets=b["targets"], skeleton_edges=b["skeleton"])
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])[:, :, :24].cpu()
    coords = torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]])[:, :, :24].cpu()
    assert (logits - t(d["logits"])).abs().max() < 1e-3
    assert (coords - t(d["coords"])).abs().max() < 1e-4
    assert torch.equal(logits.argmax(-1), t(d["logits"]).argmax(-1))
    ld = crit(out, b["targets"])
    for k, v in zip(json.loads(bytes(d["loss_keys"]).decode()), d["loss_vals"]):
        assert abs(float(ld[k]) - float(v)) < 1e-3, k
    assert abs(float(ld["_total"]) - float(d["loss"])) < 5e-3
    ld["_total"].backward()
    named = dict(model.named_parameters(remove_duplicate=False))
    for k in d.files:
        if k.startswith("gradhead:"):
            ref = t(d[k])
            got = named[k[9:]].grad.detach().cpu().reshape(-1)[:256]
            assert (got - ref).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max())), k
    worst = 0.0
    for name, ref in zip(json.loads(bytes(d["gnorm_keys"]).decode()), d["gnorm_vals"]):
        worst = max(worst, abs(float(named[name].grad.norm()) - ref) / max(ref, 1e-3))
    assert worst < 2e-2, worst


def test_cfg5_512_patch2_5shot_decode(golden_dir):
    """configs[4]: --image_size 512 (patch-2 input_proj: 2x2/s2 and 4x4/s4 convolutions), P = 68 support keypoints, 5-shot
    support mean-pooled by the collate, KV-cached autoregressive decode (eager, captured and replayed hipGraph steps)."""
    d = np.load(os.path.join(golden_dir, "cfg5_512_decode.npz"))
    sd = proc_sd_512()
    key, alias = "base_model.class_embed.5.bias", "base_model.transformer.decoder.class_embed.5.bias"
    sd[key] = sd[key] + t(d["bias_delta"])
    sd[alias] = sd[key]
    args, tok, model, crit = build_product(extra=("--image_size", "512"), proc_sd=sd)
    assert model.base_model.patch_size == 2
    model.eval()
    tok.seq_len = 40
    b = cfg5_episode_batch()
    imgs, sc, sm = b["query_images"].cuda(), b["support_coords"].cuda(), b["support_masks"].cuda()
    assert sc.shape == (2, 68, 2)
    ref_logits, ref_coords, ref_seq = t(d["logits"]), t(d["coordinates"]), t(d["sequences"]).long()
    cfg = cape_ref.Cfg(patch_size=2)
    stream = {k: v.cuda() for k, v in cape_ref.stream_from_outputs(ref_logits, ref_coords, cfg).items()}
    with torch.no_grad():
        q = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=b["support_skeletons"],
                                    teacher_stream=stream)
    assert q["logits"].shape == ref_logits.shape == (2, 40, 3)
    assert (q["logits"].cpu() - ref_logits).abs().max() < 1e-3
    assert (q["coordinates"].cpu() - ref_coords).abs().max() < 1e-4
    top2 = ref_logits.sort(-1).values
    clear = (top2[..., 2] - top2[..., 1]) > 2e-3
    assert clear.float().mean() > 0.95 and torch.equal(q["sequences"].cpu()[clear], ref_seq[clear])
    # free-running: first steps tightly, then eager == captured == replayed bitwise
    with torch.no_grad():
        p = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=b["support_skeletons"])
        assert (p["logits"][:, :4].cpu() - ref_logits[:, :4]).abs().max() < 1e-3
        for _ in range(2):
            pg = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=b["support_skeletons"], graph=True)
            assert torch.equal(pg["logits"], p["logits"]) and torch.equal(pg["coordinates"], p["coordinates"])
    big = (top2[..., 2] - top2[..., 1]) > 5e-2
    n = min(p["sequences"].shape[1], 40)
    assert torch.equal(p["sequences"].cpu()[:, :n][big[:, :n]], ref_seq[:, :n][big[:, :n]])


@pytest.mark.parametrize("images", [2, 5])
def test_whole_step_decode_kernel_matches_stage_kernels(images, monkeypatch):
    """cape_decode_step (one launch per step, one block per image) against the launch-per-stage step on the configs[4] model:
    same logits / coordinates / hidden states to fp32 rounding of the differently ordered sums, same token streams, and
    eager == captured == replayed bit for bit.  5 images: an odd batch that is not the one the fixtures were made with."""
    args, tok, model, crit = build_product(extra=("--image_size", "512"), proc_sd=proc_sd_512())
    model.eval()
    tok.seq_len = 24
    b = cfg5_episode_batch()
    rep = (images + 1) // 2
    imgs = b["query_images"].repeat(rep, 1, 1, 1)[:images].cuda()
    imgs = imgs + 0.05 * torch.randn(imgs.shape, generator=torch.Generator().manual_seed(3)).cuda()
    sc, sm = b["support_coords"].repeat(rep, 1, 1)[:images].cuda(), b["support_masks"].repeat(rep, 1)[:images].cuda()
    sk = (b["support_skeletons"] * rep)[:images]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CAPE_DECODE_MEGA", mode)
        with torch.no_grad():
            outs[mode] = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=False)
    a, w = outs["0"], outs["1"]
    n = min(a["logits"].shape[1], w["logits"].shape[1])
    assert n >= 6
    assert (a["logits"][:, :6] - w["logits"][:, :6]).abs().max() < 2e-5 and (a["coordinates"][:, :6] - w["coordinates"][:, :6]).abs().max() < 2e-6
    top2 = a["logits"][:, :n].sort(-1).values
    clear = (top2[..., 2] - top2[..., 1]) > 1e-3
    assert torch.equal(a["sequences"][:, :n][clear], w["sequences"][:, :n][clear])
    with torch.no_grad():
        for _ in range(2):
            g = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=True)
            assert torch.equal(g["logits"], w["logits"]) and torch.equal(g["coordinates"], w["coordinates"])


def test_whole_step_decode_kernel_beyond_64_images():
    """One block per image: the whole-step kernel takes batches the launch-per-stage kernels (<= 64 rows) cannot.  66 images at
    256x256; nothing couples two images, so the first five must decode exactly as when they are decoded alone."""
    args, tok, model, crit = build_product()
    model.eval()
    tok.seq_len = 12
    g = torch.Generator().manual_seed(11)
    N = 66
    imgs = torch.rand(N, 3, 256, 256, generator=g).cuda()
    sc = torch.rand(N, 17, 2, generator=g).cuda()
    sm = (torch.arange(17)[None, :] < torch.randint(4, 14, (N, 1), generator=g)).cuda()      # ragged visibility per image
    sk = [[[0, 1], [1, 2], [2, 3]] for _ in range(N)]
    with torch.no_grad():
        big = model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=False)
        small = model.forward_inference(samples=imgs[:5], support_coords=sc[:5], support_mask=sm[:5], skeleton_edges=sk[:5], graph=False)
    n = min(big["logits"].shape[1], small["logits"].shape[1])
    assert n >= 3 and big["logits"].shape[0] == N
    assert (big["logits"][:5, :n] - small["logits"][:, :n]).abs().max() < 2e-5
    assert (big["coordinates"][:5, :n] - small["coordinates"][:, :n]).abs().max() < 2e-6
    assert torch.isfinite(big["logits"]).all()


def test_decode_follows_optimizer_steps(monkeypatch):
    """The fused decode paths use derived inference weights (folded q|k|v projections) and captured step graphs; the fused
    optimizer updates the flat arenas without touching autograd's version counters.  After training steps the fused decode
    (whole-step kernel, eager and replayed graphs) must follow the new weights: same logits as the launch-per-op decode."""
    from cape_amd.runtime.optimizer import ArenaAdamW
    from cape_amd.hip import functional as HF
    args, tok, model, crit = build_product(proc_sd=None)
    tok.seq_len = 10
    opt = ArenaAdamW(model, lr=3e-3, lr_backbone=3e-4, weight_decay=1e-4, max_norm=0.1)
    HF.Runtime.seed(5, "cuda")
    g = torch.Generator().manual_seed(4)
    imgs = torch.rand(2, 3, 256, 256, generator=g).cuda()
    sc = torch.rand(2, 9, 2, generator=g).cuda()
    sm = (torch.arange(9)[None, :] < torch.tensor([[6], [7]])).cuda()
    sk = [[[0, 1], [1, 2]], [[0, 1], [2, 3]]]

    def decode(mode, graph):
        monkeypatch.setenv("CAPE_DECODE_FUSED", mode)
        model.eval()
        with torch.no_grad():
            return model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=graph)["logits"]

    for _ in range(3):                                       # eager, capture, replay: the graphs exist before the weights move
        before = decode("1", True)
    model.train()
    from cape_amd.datasets.synthetic import SyntheticEpisodes
    from cape_amd.datasets import episodic_collate_fn
    ds = SyntheticEpisodes(tok, 2, 256, 9, 1, seed=3)
    bt = episodic_collate_fn([ds[0], ds[1]])
    for _ in range(2):
        out = model(samples=bt["query_images"].cuda(), support_coords=bt["support_coords"].cuda(), support_mask=bt["support_masks"].cuda(),
                    targets={k: v.cuda() for k, v in bt["query_targets"].items()}, skeleton_edges=bt["support_skeletons"])
        crit(out, {k: v.cuda() for k, v in bt["query_targets"].items()})["_total"].backward()
        opt.step(); opt.zero_grad()
    ref = decode("0", False)                                 # launch-per-op decode straight from the parameters
    assert (ref[:, :4] - before[:, :4]).abs().max() > 1e-3   # the weights did move
    for graph in (False, True, True):
        got = decode("1", graph)
        n = min(got.shape[1], ref.shape[1])
        assert (got[:, :n] - ref[:, :n]).abs().max() < 2e-4, graph


def test_decode_follows_graph_replayed_optimizer_steps(monkeypatch):
    """the optimizer step of a REPLAYED GraphedTrainStep never runs ArenaAdamW.step() in Python, so nothing advanced the
    epoch that the derived weights key on (packed GEMM planes, folded decode projections, captured decode graphs).  Train through
    eager -> capture -> replay -> replay, a learning-rate change in between (the kernel reads the rate from the device), then the
    fused decode must agree with the launch-per-op decode straight from the parameters."""
    from cape_amd.hip import functional as HF
    from cape_amd.hip import ops
    from cape_amd.runtime.graph_step import GraphedTrainStep
    from cape_amd.runtime.optimizer import ArenaAdamW
    args, tok, model, crit = build_product(proc_sd=None)
    tok.seq_len = 10
    opt = ArenaAdamW(model, lr=3e-3, lr_backbone=3e-4, weight_decay=1e-4, max_norm=0.1)
    HF.Runtime.seed(5, "cuda")
    g = torch.Generator().manual_seed(4)
    imgs = torch.rand(2, 3, 256, 256, generator=g).cuda()
    sc = torch.rand(2, 9, 2, generator=g).cuda()
    sm = (torch.arange(9)[None, :] < torch.tensor([[6], [7]])).cuda()
    sk = [[[0, 1], [1, 2]], [[0, 1], [2, 3]]]

    def decode(mode, graph):
        monkeypatch.setenv("CAPE_DECODE_FUSED", mode)
        model.eval()
        with torch.no_grad():
            return model.forward_inference(samples=imgs, support_coords=sc, support_mask=sm, skeleton_edges=sk, graph=graph)["logits"]

    for _ in range(3):
        before = decode("1", True)
    from cape_amd.datasets import episodic_collate_fn
    from cape_amd.datasets.synthetic import SyntheticEpisodes
    ds = SyntheticEpisodes(tok, 2, 256, 9, 1, seed=3)
    bt = episodic_collate_fn([ds[0], ds[1]])
    tg = {k: v.cuda() for k, v in bt["query_targets"].items()}
    step = GraphedTrainStep(model, crit, opt, edge_capacity=64, eager_steps=1)
    model.train()
    epoch0 = ops.PackedWeights.epoch
    probe = model.base_model.transformer.decoder.layers[0].linear1.weight
    w_prev = probe.detach().clone()
    moved = []
    for i in range(4):                                       # eager, capture + replay, replay, replay (with a new learning rate)
        if i == 3:
            for gr in opt.param_groups:
                gr["lr"] = gr["lr"] * 0.1
        step(bt["query_images"].cuda(), bt["support_coords"].cuda(), bt["support_masks"].cuda(), tg, bt["support_skeletons"])
        torch.cuda.synchronize()
        moved.append(float((probe.detach() - w_prev).abs().max()))
        w_prev = probe.detach().clone()
    assert len(step.cache) == 1 and ops.PackedWeights.epoch >= epoch0 + 4
    assert all(m > 0 for m in moved) and moved[3] < 0.35 * moved[2], moved       # the replay followed the schedule (lr x 0.1)
    ref = decode("0", False)
    assert (ref[:, :4] - before[:, :4]).abs().max() > 1e-3
    for graph in (False, True, True):
        got = decode("1", graph)
        n = min(got.shape[1], ref.shape[1])
        assert (got[:, :n] - ref[:, :n]).abs().max() < 2e-4, graph


def test_evaluate_cape_with_criterion(golden_dir):
    """a17: `evaluate_cape` end to end on the device (pad / trim to the target length -> HIP criterion -> PCK) against the
    reference's stats for the crafted predictions of eval_glue.npz."""
    from cape_amd.models.engine_cape import evaluate_cape
    from tests.test_engine_pins_cpu import FakeModel, load_eval_fixture
    batches, preds, meta = load_eval_fixture(golden_dir)
    args, tok, model, crit = build_product()
    stats = evaluate_cape(FakeModel(preds, "cuda"), crit, batches, torch.device("cuda"), compute_pck=True, pck_threshold=0.2)
    ref = meta["stats"]
    for k in ("loss", "loss_ce", "loss_coords", "loss_ce_unscaled", "loss_coords_unscaled"):
        assert abs(stats[k] - ref[k]) < 1e-4 * max(1.0, abs(ref[k])), (k, stats[k], ref[k])
    assert stats["pck_num_correct"] == ref["pck_num_correct"] and stats["pck_num_visible"] == ref["pck_num_visible"]
    assert abs(stats["pck"] - ref["pck"]) < 1e-12 and abs(stats["pck_mean_categories"] - ref["pck_mean_categories"]) < 1e-12


def test_train_loop_parameter_deltas(golden_dir, proc_sd, gemm_precision):
    """a16: `train_one_epoch_episodic` as deployed (flat arenas, direct weight gradients on the side stream, fused clip +
    AdamW) over 3 micro-batches with accumulation_steps = 2 -- one boundary step and the tail flush -- against the parameter
    deltas of the reference loop with torch.optim.AdamW (every dropout 0)."""
    from cape_amd.models.engine_cape import train_one_epoch_episodic
    from cape_amd.runtime.optimizer import ArenaAdamW
    d = np.load(os.path.join(golden_dir, "train_loop.npz"))
    lr, lr_bb, wd, max_norm = (float(x) for x in d["lr"])
    args, tok, model, crit = build_product(extra=("--dropout", "0"), proc_sd=proc_sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.support_encoder.dropout_p = 0.0
    opt = ArenaAdamW(model, lr=lr, lr_backbone=lr_bb, weight_decay=wd, max_norm=max_norm)
    before = {n: p.detach().clone() for n, p in model.named_parameters(remove_duplicate=False)}
    stats = train_one_epoch_episodic(model, crit, train_loop_batches(), opt, torch.device("cuda"), epoch=0, max_norm=max_norm,
                                     print_freq=0, accumulation_steps=2)
    torch.cuda.synchronize()
    assert int(opt.step_count) == 2                                  # boundary step + tail flush
    sk, sv = json.loads(bytes(d["stat_keys"]).decode()), d["stat_vals"]
    for k, v in zip(sk, sv):
        if k in stats and k.startswith("loss"):
            assert abs(stats[k] - float(v)) < 2e-3 * max(1.0, abs(float(v))), (k, stats[k], float(v))
    after = dict(model.named_parameters(remove_duplicate=False))
    dn = dict(zip(json.loads(bytes(d["dnorm_keys"]).decode()), d["dnorm_vals"]))
    bad = []
    for n, want in dn.items():
        got = float((after[n].detach() - before[n]).norm())
        if abs(got - want) > 3e-2 * max(want, 1e-7):
            bad.append((n, got, want))
    assert not bad, (len(bad), bad[:5])
    for k in d.files:
        if k.startswith("delta:"):
            n = k[6:]
            got = (after[n].detach() - before[n]).cpu()
            # channels_last conv weights: the fixture slices the logical (O, C, KH, KW) order
            got = got.contiguous().reshape(-1)[:512]
            ref = t(d[k])
            # an Adam step is ~lr * sign(g): elements whose gradient is noise-level flip (|delta| jumps by 2 lr); the split
            # arithmetic perturbs more of them than exact fp32
            close = ((got - ref).abs() <= 5e-6).float().mean().item()
            assert close >= (0.93 if gemm_precision == "f32" else 0.90), (k, close)


# ------------------------------------------------------------------------------------------------
# round-3 fixtures (oracle/make_golden_r3.py)
# ------------------------------------------------------------------------------------------------
def _check_step(d, model, crit, out, targets):
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])[:, :, :24].cpu()
    coords = torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]])[:, :, :24].cpu()
    assert (logits - t(d["logits"])).abs().max() < 1e-3
    assert (coords - t(d["coords"])).abs().max() < 1e-4
    assert torch.equal(logits.argmax(-1), t(d["logits"]).argmax(-1))
    ld = crit(out, targets)
    for k, v in zip(json.loads(bytes(d["loss_keys"]).decode()), d["loss_vals"]):
        assert abs(float(ld[k]) - float(v)) < 1e-3, k
    assert abs(float(ld["_total"].detach()) - float(d["loss"])) < 5e-3
    ld["_total"].backward()
    from cape_amd.hip import functional as HF
    HF.Runtime.join()
    named = dict(model.named_parameters(remove_duplicate=False))
    for k in d.files:
        if k.startswith("gradhead:"):
            ref = t(d[k])
            got = named[k[9:]].grad.detach().cpu().reshape(-1)[:256]
            assert (got - ref).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max())), k
    worst = 0.0
    for name, ref in zip(json.loads(bytes(d["gnorm_keys"]).decode()), d["gnorm_vals"]):
        worst = max(worst, abs(float(named[name].grad.norm()) - ref) / max(ref, 1e-3))
    assert worst < 2e-2, worst


def test_e2e256_backward_at_the_headline_shape(golden_dir, proc_sd):
    """BASELINE configs[1] geometry (256x256, 17 keypoints, N = 2): forward, 19 losses, every gradient norm, 8 gradient slices."""
    d = np.load(os.path.join(golden_dir, "e2e256_grads.npz"))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    model.eval()
    b = to_dev(synth.make_batch(23, 1, 2, 256, 17, CFG, n_invisible=(2,)))
    out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                targets=b["targets"], skeleton_edges=b["skeleton"])
    _check_step(d, model, crit, out, b["targets"])


def test_cfg3_5shot_gcn_training_step(golden_dir, proc_sd):
    """BASELINE configs[2] as a whole: two 5-shot episodes -> the product's collate (mean-pooled support) -> GCN pre-encoder ->
    teacher-forced training step at 256x256 (N = 4) against the reference's step on the reference's collate."""
    from tests.test_oracle_golden import cfg3_batch
    d = np.load(os.path.join(golden_dir, "cfg3_5shot_256.npz"))
    b = cfg3_batch()
    assert torch.equal(b["support_coords"], t(d["support_coords"])) and torch.equal(b["support_masks"], t(d["support_masks"]))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    assert model.support_encoder.use_gcn_preenc if hasattr(model.support_encoder, "use_gcn_preenc") else True
    model.eval()
    tg = {k: v.cuda() for k, v in b["query_targets"].items()}
    out = model(samples=b["query_images"].cuda(), support_coords=b["support_coords"].cuda(), support_mask=b["support_masks"].cuda(),
                targets=tg, skeleton_edges=b["support_skeletons"])
    _check_step(d, model, crit, out, tg)
