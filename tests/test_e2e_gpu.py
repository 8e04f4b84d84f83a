"""End-to-end GPU parity: the HIP product path (CAPEModel on libcape_hip.so) against
(a) golden vectors emitted by the real reference and (b) the CPU oracle on the same seeded inputs.
Tolerance from the north star: logits within 1e-3, argmax tokens exact (where the reference's top-2
margin is clear), coordinates within 1e-4."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cape_ref, synth
from tests.helpers import build_product, to_dev

CFG = cape_ref.Cfg()


@pytest.fixture(params=["bf16x3", "f32"], autouse=True)
def gemm_precision(request):
    """Every end-to-end parity test runs in both GEMM arithmetic modes (default bf16x3 split, exact fp32)."""
    from cape_amd.hip import ops
    old = ops.get_gemm_precision()
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision(old)


def t(a):
    return torch.from_numpy(np.asarray(a))


def stack_outputs(out):
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])
    coords = torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]])
    return logits, coords


def test_state_dict_keys_match_reference(proc_sd):
    _, _, model, _ = build_product(device="cpu")
    sd = model.state_dict()
    assert list(sd.keys()) == list(proc_sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(proc_sd[k].shape), k
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_train == 47973876 and sum(p.numel() for p in model.parameters()) == 48247476


def test_forward_loss_grads_64_vs_reference_golden(golden_dir, proc_sd):
    d = np.load(os.path.join(golden_dir, "e2e64.npz"))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    model.eval()
    b = to_dev(synth.make_batch(11, 2, 2, 64, 9, CFG, n_invisible=(2, 0)))
    out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                targets=b["targets"], skeleton_edges=b["skeleton"])
    logits, coords = stack_outputs(out)
    assert (logits.cpu() - t(d["logits"])).abs().max() < 1e-3
    assert (coords.cpu() - t(d["coords"])).abs().max() < 1e-4
    assert (out["pred_room_logits"][:, :16].cpu() - t(d["room_logits"])).abs().max() < 1e-3
    assert torch.equal(logits.argmax(-1).cpu(), t(d["logits"]).argmax(-1))
    ld = crit(out, b["targets"])
    keys = json.loads(bytes(d["loss_keys"]).decode())
    assert sorted(k for k in ld if not k.startswith("_")) == keys
    for k, v in zip(keys, d["loss_vals"]):
        assert abs(float(ld[k]) - float(v)) < 1e-3, k
    total = ld["_total"]
    assert abs(float(total) - float(d["loss"])) < 5e-3
    total.backward()
    named = dict(model.named_parameters(remove_duplicate=False))
    for k in d.files:
        if k.startswith("grad:") or k.startswith("gradhead:"):
            name = k.split(":", 1)[1]
            ref = t(d[k])
            got = named[name].grad.detach().cpu()
            got = got.reshape(-1)[:256] if k.startswith("gradhead:") else got
            tol = 2e-3 * max(1.0, float(ref.abs().max()))
            assert (got.reshape(ref.shape) - ref).abs().max() <= tol, (k, float((got.reshape(ref.shape) - ref).abs().max()))
    gk = json.loads(bytes(d["gnorm_keys"]).decode())
    worst = 0.0
    for name, ref in zip(gk, d["gnorm_vals"]):
        got = float(named[name].grad.norm())
        worst = max(worst, abs(got - ref) / max(ref, 1e-3))
    assert worst < 2e-2, worst
    dead = json.loads(bytes(d["no_grad_names"]).decode())
    for name in dead:
        g = named[name].grad
        assert g is None or float(g.abs().sum()) == 0.0, name


def test_forward_256_vs_reference_golden(golden_dir, proc_sd):
    d = np.load(os.path.join(golden_dir, "e2e256.npz"))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    model.eval()
    b = to_dev(synth.make_batch(23, 1, 2, 256, 17, CFG, n_invisible=(2,)))
    with torch.no_grad():
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                    targets=b["targets"], skeleton_edges=b["skeleton"])
        ld = crit(out, b["targets"])
    logits, coords = stack_outputs(out)
    assert (logits[:, :, :24].cpu() - t(d["logits"])).abs().max() < 1e-3
    assert (coords[:, :, :24].cpu() - t(d["coords"])).abs().max() < 1e-4
    keys = json.loads(bytes(d["loss_keys"]).decode())
    for k, v in zip(keys, d["loss_vals"]):
        assert abs(float(ld[k]) - float(v)) < 1e-3, k


@pytest.mark.parametrize("name", ["e2e64_decode.npz", "e2e64_decode_eos.npz"])
def test_cached_decode_vs_reference_golden(golden_dir, proc_sd, name):
    d = np.load(os.path.join(golden_dir, name))
    sd = dict(proc_sd)
    key = "base_model.class_embed.5.bias"
    alias = "base_model.transformer.decoder.class_embed.5.bias"
    if "bias_delta" in d.files:
        sd[key] = sd[key] + t(d["bias_delta"]); max_len = 40
    else:
        sd[key] = t(d["bias"]); max_len = 200
    sd[alias] = sd[key]
    args, tok, model, crit = build_product(proc_sd=sd)
    model.eval()
    tok.seq_len = max_len
    b = to_dev(synth.make_batch(11, 2, 2, 64, 9, CFG, n_invisible=(2, 0)))
    ref_logits, ref_coords = t(d["logits"]), t(d["coordinates"])
    with torch.no_grad():
        p = model.forward_inference(samples=b["images"], support_coords=b["support_coords"],
                                    support_mask=b["support_mask"], skeleton_edges=b["skeleton"])
    assert p["logits"].shape == ref_logits.shape, (p["logits"].shape, ref_logits.shape)
    assert (p["logits"][:, :4].cpu() - ref_logits[:, :4]).abs().max() < 1e-3
    # hipGraph path: call 1 above was eager, call 2 captures one graph per step while decoding, call 3 replays them;
    # same kernels in the same order -> bitwise the eager result
    with torch.no_grad():
        for _ in range(2):
            pg = model.forward_inference(samples=b["images"], support_coords=b["support_coords"],
                                         support_mask=b["support_mask"], skeleton_edges=b["skeleton"], graph=True)
            assert torch.equal(pg["logits"], p["logits"]) and torch.equal(pg["coordinates"], p["coordinates"])
        pe = model.forward_inference(samples=b["images"], support_coords=b["support_coords"],
                                     support_mask=b["support_mask"], skeleton_edges=b["skeleton"], graph=False)
        assert torch.equal(pe["logits"], p["logits"])
    assert len(model.base_model._decode_states) == 1 and len(next(iter(model.base_model._decode_states.values()))["graphs"]) >= p["logits"].shape[1]
    top2 = ref_logits.sort(-1).values
    clear = (top2[..., 2] - top2[..., 1]) > 5e-2
    assert torch.equal(p["sequences"].cpu()[clear], t(d["sequences"]).long()[clear])
    # teacher-forced on the reference's own stream: every step within tolerance, tokens exact
    stream = {k: v.to("cuda") for k, v in cape_ref.stream_from_outputs(ref_logits, ref_coords, CFG).items()}
    with torch.no_grad():
        q = model.forward_inference(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                                    skeleton_edges=b["skeleton"], teacher_stream=stream)
    assert (q["logits"].cpu() - ref_logits).abs().max() < 1e-3
    assert (q["coordinates"].cpu() - ref_coords).abs().max() < 1e-4
    assert torch.equal(q["sequences"].cpu(), t(d["sequences"]).long())


def test_train_mode_step_runs_and_is_finite(proc_sd):
    """Dropout-on training forward/backward (bit parity is impossible with dropout): finite loss and grads,
    loss close to the eval-mode value."""
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    b = to_dev(synth.make_batch(3, 2, 2, 64, 9, CFG, n_invisible=(2, 0)))
    model.eval()
    with torch.no_grad():
        l_eval = float(crit(model(b["images"], b["support_coords"], b["support_mask"], b["targets"], b["skeleton"]), b["targets"])["_total"])
    model.train()
    out = model(b["images"], b["support_coords"], b["support_mask"], b["targets"], b["skeleton"])
    total = crit(out, b["targets"])["_total"]
    total.backward()
    assert torch.isfinite(total) and abs(float(total) - l_eval) < 0.5 * abs(l_eval)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n


def test_arena_direct_grads_and_optimizer_step(golden_dir, proc_sd):
    """Training path as deployed: flat arenas, wgrad kernels accumulating straight into the gradient arena on
    the side stream, fused clip + AdamW.  Gradients against the reference goldens; the update against
    torch.optim.AdamW + clip_grad_norm_ fed with the same gradients on the CPU."""
    import cape_amd  # noqa: F401
    from cape_amd.runtime.optimizer import ArenaAdamW
    d = np.load(os.path.join(golden_dir, "e2e64.npz"))
    args, tok, model, crit = build_product(proc_sd=proc_sd)
    model.eval()                                           # dropout off -> comparable with the goldens
    opt = ArenaAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, max_norm=0.1)
    names = {id(p): n for n, p in model.named_parameters()}
    b = to_dev(synth.make_batch(11, 2, 2, 64, 9, CFG, n_invisible=(2, 0)))
    for rep in range(2):                                   # second pass checks zero_grad + re-accumulation
        opt.zero_grad()
        out = model(samples=b["images"], support_coords=b["support_coords"], support_mask=b["support_mask"],
                    targets=b["targets"], skeleton_edges=b["skeleton"])
        crit(out, b["targets"])["_total"].backward()
        torch.cuda.synchronize()
        named = dict(model.named_parameters(remove_duplicate=False))
        gk = json.loads(bytes(d["gnorm_keys"]).decode())
        worst = 0.0
        for name, ref in zip(gk, d["gnorm_vals"]):
            worst = max(worst, abs(float(named[name].grad.norm()) - ref) / max(ref, 1e-3))
        assert worst < 2e-2, (rep, worst)
        for k in d.files:
            if k.startswith("grad:"):
                ref = t(d[k])
                got = named[k[5:]].grad.detach().cpu()
                assert (got - ref).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max())), (rep, k)
    # optimizer step vs torch
    cpu_params, cpu_groups = [], [[], []]
    for gi, a in enumerate(opt.arenas):
        for p in a.params:
            q = torch.nn.Parameter(p.detach().cpu().clone())
            q.grad = p.grad.detach().cpu().clone()
            cpu_groups[gi].append(q)
    ref_opt = torch.optim.AdamW([{"params": cpu_groups[0]}, {"params": cpu_groups[1], "lr": 1e-5}], lr=1e-4, weight_decay=1e-4)
    total_norm = torch.nn.utils.clip_grad_norm_(cpu_groups[0] + cpu_groups[1], 0.1)
    ref_opt.step()
    opt.step()
    torch.cuda.synchronize()
    assert abs(float(opt.grad_norm()) - float(total_norm)) <= 1e-3 * float(total_norm)
    for gi, a in enumerate(opt.arenas):
        for p, q in zip(a.params, cpu_groups[gi]):
            assert (p.detach().cpu() - q.detach()).abs().max() <= 1e-6, names[id(p)]
    for n, p in opt.dead:                                   # never-grad parameters are untouched
        assert torch.equal(p.detach().cpu(), proc_sd[n])


def test_graphed_train_step_matches_eager(monkeypatch):
    """runtime/graph_step.py: replaying the captured step gives the eager step's losses and parameters (same RNG stream).
    Two executions of the same mathematics are compared, so the forward pass runs without atomic k-splits (CAPE_DETERMINISTIC:
    the model turns their 1e-7 arrival-order rounding into 5e-3 of the gradient, profiles/r02_determinism.txt)."""
    import copy
    from cape_amd.hip import functional as _HF
    monkeypatch.setattr(_HF, "_DETERMINISTIC", True)
    from cape_amd.runtime.graph_step import GraphedTrainStep
    from cape_amd.runtime.optimizer import ArenaAdamW
    from cape_amd.hip import functional as HF
    from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn
    from cape_amd.datasets.synthetic import SyntheticEpisodes
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    import argparse
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", "64"])
    tok = DiscreteTokenizerV2(44, args.seq_len)
    DEV = "cuda"
    ds = SyntheticEpisodes(tok, 12, 64, 17, 2, seed=5)
    batches = []
    for i in range(3):
        b = episodic_collate_fn([ds[i * 4 + j] for j in range(4)])
        batches.append((b["query_images"].to(DEV), b["support_coords"].to(DEV), b["support_masks"].to(DEV),
                        {k: v.to(DEV) for k, v in b["query_targets"].items()}, b["support_skeletons"]))

    def run(graphed):
        from cape_amd.hip import ops
        torch.manual_seed(0)
        ops._stream_counter[0] = 0                            # dropout stream ids are handed out at construction
        base, crit = build_model(args, tokenizer=tok)
        model = build_cape_model(args, base).to(DEV)
        crit = crit.to(DEV)
        model.train()
        HF.Runtime.seed(77, DEV)
        opt = ArenaAdamW(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, max_norm=0.1)
        step = GraphedTrainStep(model, crit, opt, edge_capacity=512, eager_steps=(1 if graphed else 10 ** 9))
        losses = []
        for it in range(5):                                   # call 0 eager, call 1 captures + replays, calls 2.. replay
            im, sc, sm, tg, sk = batches[it % 3]
            losses.append(float(step(im, sc, sm, tg, sk)["_total"]))
        assert (len(step.cache) == 1) == graphed
        flat = torch.cat([p.detach().reshape(-1)[:64] for p in model.parameters() if p.requires_grad][:40])
        return losses, flat

    le, pe = run(False)
    lg, pg = run(True)
    for a_, b_ in zip(le, lg):
        assert abs(a_ - b_) <= 2e-4 * max(1.0, abs(a_)), (le, lg)
    # reduction order differs from run to run (split-K / LDS atomics) and Adam turns a noise-level gradient into a +-lr move:
    # bound the worst element by steps x lr and ask the bulk to agree closely
    d = (pe - pg).abs()
    assert d.max().item() <= 6e-4 and d.mean().item() <= 5e-6, (d.max().item(), d.mean().item())


def test_bixattn_blocks_against_reference_golden(golden_dir):
    """the standalone bidirectional attention blocks against the reference's own classes."""
    from oracle import procweights
    from oracle.make_golden_bixattn import inputs
    from cape_amd.models.bixattn import BiXAttnBlock, CAOneSidedBlock
    d = np.load(os.path.join(golden_dir, "bixattn.npz"))
    lat, pat = (x.cuda() for x in inputs())

    def fill(mod, prefix):
        mod.load_state_dict({k: procweights.tensor_for(prefix + "." + k, tuple(v.shape)) for k, v in mod.state_dict().items()}, strict=True)
        return mod.cuda().eval()

    tol = 2e-5 if _precision() == "f32" else 2e-4
    ol, op = fill(BiXAttnBlock(256, 256, 256, 8, init_values=0.1), "bixattn.bi")(lat, pat)
    assert (ol.cpu() - t(d["bi_lat"])).abs().max() <= tol and (op.cpu()[:, ::5] - t(d["bi_pat"])).abs().max() <= tol
    ol, op = fill(BiXAttnBlock(256, 256, 256, 8, rv_bias=True, init_values=None), "bixattn.bi0")(lat, pat)
    assert (ol.cpu() - t(d["bi0_lat"])).abs().max() <= 4 * tol and (op.cpu()[:, ::5] - t(d["bi0_pat"])).abs().max() <= 4 * tol
    oo, none = fill(CAOneSidedBlock(256, 256, 256, 8, init_values=0.1), "bixattn.one")(lat, pat)
    assert none is None and (oo.cpu() - t(d["one_lat"])).abs().max() <= tol
    with pytest.raises(RuntimeError):                       # an active drop rate has no kernel on this path
        BiXAttnBlock(256, 256, 256, 8, drop=0.1).cuda().train()(lat, pat)


def test_bixattn_blocks_backward_against_reference_golden(golden_dir):
    """Row a14, backward: input gradients and every parameter gradient (norm + first 8 elements) of the three blocks against the
    reference's own classes under autograd (oracle/make_golden_bixattn.py::grads; train mode, all drop rates 0)."""
    from oracle import procweights
    from oracle.make_golden_bixattn import inputs
    from cape_amd.models.bixattn import BiXAttnBlock, CAOneSidedBlock
    d = np.load(os.path.join(golden_dir, "bixattn_grads.npz"))
    rng = np.random.Generator(np.random.PCG64(32))
    c_lat = torch.from_numpy(rng.standard_normal((2, 24, 256)).astype(np.float32)).cuda()
    c_pat = torch.from_numpy(rng.standard_normal((2, 280, 256)).astype(np.float32)).cuda()
    lat0, pat0 = inputs()

    def fill(mod, prefix):
        mod.load_state_dict({k: procweights.tensor_for(prefix + "." + k, tuple(v.shape)) for k, v in mod.state_dict().items()}, strict=True)
        return mod.cuda().train()

    rel = 2e-5 if _precision() == "f32" else 3e-4
    for name, mod in (("bi", fill(BiXAttnBlock(256, 256, 256, 8, init_values=0.1), "bixattn.bi")),
                      ("bi0", fill(BiXAttnBlock(256, 256, 256, 8, rv_bias=True, init_values=None), "bixattn.bi0")),
                      ("one", fill(CAOneSidedBlock(256, 256, 256, 8, init_values=0.1), "bixattn.one"))):
        xl, xp = lat0.cuda().requires_grad_(True), pat0.cuda().requires_grad_(True)
        ol, op = mod(xl, xp)
        loss = (ol * c_lat).sum() + ((op * c_pat).sum() if op is not None else 0.0)
        loss.backward()
        for got, key in ((xl.grad, name + "_dlat"), (xp.grad[:, ::5], name + "_dpat")):
            ref = t(d[key])
            assert (got.cpu() - ref).abs().max() <= rel * max(1.0, float(ref.abs().max())), (name, key)
        grads = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
        assert sorted(grads) == sorted(str(k) for k in d[name + "_pnames"])
        for k, norm, head in zip(d[name + "_pnames"], d[name + "_pnorms"], d[name + "_pheads"]):
            g = grads[str(k)]
            assert abs(float(g.norm()) - norm) <= 5 * rel * max(norm, 1e-3), (name, k, float(g.norm()), norm)
            assert (g.reshape(-1)[:8].cpu() - t(head)).abs().max() <= 5 * rel * max(1.0, float(g.abs().max())), (name, k)


def _precision():
    from cape_amd.hip import ops
    return ops.get_gemm_precision()
