"""TEST INFRASTRUCTURE ONLY -- emits tests/golden/* from the REAL reference.

Runs only in the build container (needs /root/reference, imported in place through
oracle/refshim.py; nothing is copied).  Fixtures are data: seeds/inputs and the
reference's outputs.  Re-run:  python -m oracle.make_golden
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import refshim, procweights, synth, cape_ref  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
warnings.filterwarnings("ignore")


def npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **conv)
    print("wrote", name, {k: v.shape for k, v in conv.items()})


def ref_tokenizer(tok):
    return lambda k, h, w, v, c: refshim.reference_tokenize(tok, k, h, w, v, c)


def load_procedural(model):
    built = {k: v for k, v in model.state_dict().items() if k in procweights.KEEP_AS_BUILT}
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = procweights.procedural_state_dict(spec, built)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    return spec


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    args, tok, model, crit = refshim.build_reference()
    cfg = cape_ref.Cfg()
    spec = load_procedural(model)
    with open(os.path.join(OUT, "state_dict_spec.json"), "w") as f:
        json.dump([[k, list(s)] for k, s in spec], f)
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    with open(os.path.join(OUT, "trainable_names.json"), "w") as f:
        json.dump(trainable, f)
    # as-built deterministic buffers: check the formulas in procweights reproduce them
    sd0 = model.state_dict()
    for k in procweights.KEEP_AS_BUILT:
        assert torch.equal(procweights._as_built(k, tuple(sd0[k].shape)), sd0[k]) or \
            torch.allclose(procweights._as_built(k, tuple(sd0[k].shape)), sd0[k], atol=1e-7), k

    # ---------------- 1. tokenizer (datasets/mp100_cape.py:625-832) ----------------
    rng = np.random.Generator(np.random.PCG64(7))
    kp = rng.random((9, 2)) * 256
    kp[3] = [256.0, 0.0]          # boundary: exactly 1.0 / 0.0
    kp[4] = [300.0, -5.0]         # out of range -> clamped
    vis = [2, 2, 0, 1, 2, 0, 2, 2, 2]
    t = refshim.reference_tokenize(tok, [tuple(p) for p in kp], 256, 256, vis, 5)
    npz("tokenizer.npz", kpts=kp, vis=np.array(vis), **{k: v for k, v in t.items()})

    # ---------------- 2. adjacency + support encoder ----------------
    from models.graph_utils import adj_from_skeleton
    P = 7
    coords = torch.from_numpy(rng.random((4, P, 2), dtype=np.float32))
    enc_mask = torch.tensor([[0, 1, 0, 1, 1, 0, 0],      # not left aligned
                             [1, 1, 1, 1, 1, 1, 1],      # all masked -> zeros
                             [0, 0, 0, 1, 1, 1, 1],      # left aligned
                             [0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    skel = [[[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [5, 6]], [[0, 1], [1, 2], [5, 9]], [[0, 2], [2, 1], [1, 0], [3, 4]], []]
    adj = adj_from_skeleton(P, skel, enc_mask, "cpu")
    se = model.support_encoder
    model.eval()
    out_grad = se(coords, enc_mask, skel)                         # eval, grad enabled -> slow path
    with torch.no_grad():
        out_nograd = se(coords, enc_mask, skel)                   # not left aligned over batch -> slow path
        out_fast = se(coords[2:], enc_mask[2:], skel[2:])         # left aligned batch -> nested fast path
        out_allm = se(coords[1:2], enc_mask[1:2], skel[1:2])
    npz("support_encoder.npz", coords=coords, enc_mask=enc_mask, adj=adj, out_grad=out_grad,
        out_nograd=out_nograd, out_fast=out_fast, out_allmasked=out_allm,
        skel_json=np.frombuffer(json.dumps(skel).encode(), dtype=np.uint8))

    # ---------------- 3. MSDA core (deformable_transformer.py:115-141) ----------------
    from models.deformable_transformer import ms_deform_attn_core_pytorch
    shapes = [(8, 8), (4, 4), (2, 2), (1, 1)]
    S = sum(h * w for h, w in shapes)
    value = torch.from_numpy(rng.standard_normal((1, S, 8, 32)).astype(np.float32))
    loc = torch.from_numpy(rng.uniform(-0.2, 1.2, (1, 7, 8, 4, 4, 2)).astype(np.float32))
    aw = torch.softmax(torch.from_numpy(rng.standard_normal((1, 7, 8, 16)).astype(np.float32)), -1).view(1, 7, 8, 4, 4)
    value.requires_grad_(True); loc.requires_grad_(True); aw.requires_grad_(True)
    o = ms_deform_attn_core_pytorch(value, torch.tensor(shapes), loc, aw)
    gout = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
    o.backward(gout)
    npz("msda_core.npz", shapes=np.array(shapes), value=value, loc=loc, aw=aw, out=o, gout=gout,
        g_value=value.grad, g_loc=loc.grad, g_aw=aw.grad)

    # ---------------- 4. end-to-end teacher-forced forward + loss + grads, 64x64 ----------------
    for tag, R, B in (("e2e64", 64, 2),):
        batch = synth.make_batch(11, B, 2, R, 9, cfg, n_invisible=(2, 0), tokenizer=ref_tokenizer(tok))
        # oracle tokenizer == reference tokenizer on these inputs
        b2 = synth.make_batch(11, B, 2, R, 9, cfg, n_invisible=(2, 0))
        for k in batch["targets"]:
            assert torch.equal(batch["targets"][k], b2["targets"][k]), k
        model.eval()
        model.zero_grad(set_to_none=True)
        out = model(samples=batch["images"], support_coords=batch["support_coords"],
                    support_mask=batch["support_mask"], targets=batch["targets"], skeleton_edges=batch["skeleton"])
        ld = crit(out, batch["targets"])
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        loss.backward()
        grads = {}
        named = dict(model.named_parameters(remove_duplicate=False))
        gnorm_all = {}
        for n, p in named.items():
            if p.grad is not None:
                gnorm_all[n] = float(p.grad.norm())
        picks = ["base_model.class_embed.5.weight", "base_model.query_embed.weight", "base_model.transformer.level_embed",
                 "base_model.coords_embed.0.layers.2.weight", "support_encoder.coord_mlp.0.weight",
                 "support_encoder.gcn_layers.0.conv.bias", "base_model.transformer.decoder.pos_trans_norm.weight",
                 "base_model.input_proj.3.1.weight", "base_model.input_proj.0.0.bias"]
        for n in picks:
            grads["grad:" + n] = named[n].grad
        head = {}
        for n in ["base_model.backbone.0.body.layer2.0.conv1.weight", "base_model.backbone.0.body.layer4.2.conv2.weight",
                  "base_model.transformer.encoder.layers.0.linear1.weight",
                  "base_model.transformer.encoder.layers.5.self_attn.sampling_offsets.weight",
                  "base_model.transformer.decoder.layers.0.attn_q.weight",
                  "base_model.transformer.decoder.layers.3.self_attn.in_proj_weight",
                  "base_model.transformer.decoder.layers.5.support_attn.out_proj.weight",
                  "base_model.transformer.decoder.token_embed.weight", "base_model.input_proj.3.0.weight"]:
            head["gradhead:" + n] = named[n].grad.reshape(-1)[:256]
        no_grad_names = sorted(n for n, p in named.items() if p.requires_grad and p.grad is None)
        npz(tag + ".npz",
            logits=torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]),
            coords=torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]]),
            room_logits=out["pred_room_logits"][:, :16], loss=loss,
            loss_keys=np.frombuffer(json.dumps(sorted(ld.keys())).encode(), dtype=np.uint8),
            loss_vals=np.array([float(ld[k]) for k in sorted(ld.keys())]),
            gnorm_keys=np.frombuffer(json.dumps(sorted(gnorm_all)).encode(), dtype=np.uint8),
            gnorm_vals=np.array([gnorm_all[k] for k in sorted(gnorm_all)]),
            no_grad_names=np.frombuffer(json.dumps(no_grad_names).encode(), dtype=np.uint8),
            **grads, **head)

        # ---------------- 5. cached AR decode on the same episodes ----------------
        # the procedural class head always says <eos>; shift its bias so that the stream mixes
        # <coord>/<sep>/<eos> (exercises every branch of roomformer_v2.py:530-597)
        delta = torch.tensor([2.2, 1.9, 0.0])
        with torch.no_grad():
            model.base_model.class_embed[5].bias.add_(delta)
        tok.seq_len = 40
        with torch.no_grad():
            pred = model.forward_inference(samples=batch["images"], support_coords=batch["support_coords"],
                                           support_mask=batch["support_mask"], skeleton_edges=batch["skeleton"])
        tok.seq_len = 200
        with torch.no_grad():
            model.base_model.class_embed[5].bias.sub_(delta)
        npz(tag + "_decode.npz", logits=pred["logits"], coordinates=pred["coordinates"], sequences=pred["sequences"],
            bias_delta=delta)

    # EOS / min_len bookkeeping: force EOS by a crafted head bias
    sd_eos = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model.base_model.class_embed[5].bias.copy_(torch.tensor([0.0, -5.0, 9.0]))
    batch = synth.make_batch(11, 2, 2, 64, 9, cfg, n_invisible=(2, 0))
    with torch.no_grad():
        pred = model.forward_inference(samples=batch["images"], support_coords=batch["support_coords"],
                                       support_mask=batch["support_mask"], skeleton_edges=batch["skeleton"])
    npz("e2e64_decode_eos.npz", logits=pred["logits"], coordinates=pred["coordinates"], sequences=pred["sequences"],
        bias=np.array([0.0, -5.0, 9.0], dtype=np.float32))
    model.load_state_dict(sd_eos)

    # ---------------- 6. 256x256 forward (N=2), outputs only ----------------
    batch = synth.make_batch(23, 1, 2, 256, 17, cfg, n_invisible=(2,))
    with torch.no_grad():
        model.eval()
        out = model(samples=batch["images"], support_coords=batch["support_coords"],
                    support_mask=batch["support_mask"], targets=batch["targets"], skeleton_edges=batch["skeleton"])
        ld = crit(out, batch["targets"])
    npz("e2e256.npz",
        logits=torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])[:, :, :24],
        coords=torch.stack([a["pred_coords"] for a in out["aux_outputs"]] + [out["pred_coords"]])[:, :, :24],
        loss_keys=np.frombuffer(json.dumps(sorted(ld.keys())).encode(), dtype=np.uint8),
        loss_vals=np.array([float(ld[k]) for k in sorted(ld.keys())]))

    # ---------------- 7. PCK known answers through the reference metric ----------------
    from util.eval_utils import compute_pck_bbox
    pk = rng.random((6, 2)) * 512
    gk = pk + rng.normal(0, 40, (6, 2))
    v = np.array([2, 1, 0, 2, 2, 0])
    r = compute_pck_bbox(pk, gk, 200.0, 150.0, v, 0.2)
    npz("pck.npz", pred=pk, gt=gk, vis=v, bbox=np.array([200.0, 150.0]), result=np.array(r, dtype=np.float64))


if __name__ == "__main__":
    main()
