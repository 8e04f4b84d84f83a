"""TEST INFRASTRUCTURE ONLY -- round-2 golden vectors from the REAL reference (build container only;
/root/reference is imported in place through oracle/refshim.py, nothing is copied).  Fixtures are data: seeds /
inputs and the reference's outputs.  Re-run:  python -m oracle.make_golden_r2 [cfg4] [cfg5] [eval] [train]

  cfg4_384.npz          BASELINE configs[3] minus Swin-T: ResNet-50 at 384x384 (S = 3060 tokens), N = 2 query images,
                        17 keypoints: 6-layer logits / coords (first 24 positions), the 19 losses, selected gradient
                        norms and slices of the teacher-forced step.
  cfg5_512_decode.npz   BASELINE configs[4]: --image_size 512 (patch-2 input_proj, roomformer_v2.py:995), P = 68 support
                        keypoints, 5-shot support mean-pooled by the reference's own episodic_collate_fn, KV-cached
                        autoregressive decode for 40 steps.  state_dict_spec_512_diff.json = the tensors whose shape
                        differs from the 256 model.
  eval_glue.npz/.json   `evaluate_cape` (engine_cape.py:394-870) on crafted predictions: early EOS (zero padding), excess
                        keypoints (trim), a <sep> in the stream, ragged categories, T < L and T = L, a batch without
                        query_metadata -> pck, per-batch counters, validation losses.
  train_loop.npz        `train_one_epoch_episodic` (engine_cape.py:48-301), 3 micro-batches, accumulation_steps = 2
                        (one boundary step + the tail flush), clip 0.1, AdamW, every dropout set to 0: parameter deltas.
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
from oracle.make_golden import OUT, load_procedural, npz, ref_tokenizer  # noqa: E402

os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
warnings.filterwarnings("ignore")


def jbytes(obj):
    return np.frombuffer(json.dumps(obj).encode(), dtype=np.uint8)


def stack_layers(out, key):
    return torch.stack([a[key] for a in out["aux_outputs"]] + [out[key]])


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


# --------------------------------------------------------------------------------------------------------------
def cfg4():
    args, tok, model, crit = refshim.build_reference()
    cfg = cape_ref.Cfg()
    load_procedural(model)
    model.eval()
    batch = synth.make_batch(31, 1, 2, 384, 17, cfg, n_invisible=(2,), tokenizer=ref_tokenizer(tok))
    model.zero_grad(set_to_none=True)
    out = model(samples=batch["images"], support_coords=batch["support_coords"], support_mask=batch["support_mask"],
                targets=batch["targets"], skeleton_edges=batch["skeleton"])
    ld = crit(out, batch["targets"])
    loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    loss.backward()
    named = dict(model.named_parameters(remove_duplicate=False))
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    heads = {}
    for n in ["base_model.backbone.0.body.layer3.2.conv2.weight", "base_model.input_proj.1.0.weight",
              "base_model.transformer.encoder.layers.2.linear2.weight",
              "base_model.transformer.decoder.layers.4.cross_attn.value_proj.weight",
              "base_model.transformer.decoder.layers.1.support_attn.in_proj_weight", "support_encoder.coord_mlp.2.weight"]:
        heads["gradhead:" + n] = named[n].grad.reshape(-1)[:256]
    npz("cfg4_384.npz", logits=stack_layers(out, "pred_logits")[:, :, :24], coords=stack_layers(out, "pred_coords")[:, :, :24],
        loss=loss, loss_keys=jbytes(sorted(ld.keys())), loss_vals=np.array([float(ld[k]) for k in sorted(ld.keys())]),
        gnorm_keys=jbytes(sorted(gn)), gnorm_vals=np.array([gn[k] for k in sorted(gn)]), **heads)


# --------------------------------------------------------------------------------------------------------------
def cfg5():
    args, tok, model, crit = refshim.build_reference(["--image_size", "512"])
    assert model.base_model.patch_size == 2
    cfg = cape_ref.Cfg(patch_size=2)
    spec = load_procedural(model)
    base = dict(procweights.load_spec())
    diff = [[k, list(s)] for k, s in spec if base.get(k) != tuple(s)]
    assert len(spec) == len(base) and len(diff) <= 8, (len(spec), len(base), len(diff))
    with open(os.path.join(OUT, "state_dict_spec_512_diff.json"), "w") as f:
        json.dump(diff, f)
    print("spec diff", diff)
    from datasets.episodic_sampler import episodic_collate_fn as ref_collate
    ep = synth.make_episode(41, 512, 68, 2, 5, cfg, tokenizer=ref_tokenizer(tok))
    b = ref_collate([ep])
    assert b["support_coords"].shape == (2, 68, 2)
    # shift the class head so that the stream mixes <coord> / <sep> / <eos> and still runs >= 32 steps (the procedural head
    # of this geometry says <eos>-vs-<coord> with ~0.1 margins and never <sep>): first candidate that does both
    model.eval()
    bias0 = model.base_model.class_embed[5].bias.detach().clone()
    pred, delta = None, None
    for cand in ([0.3, 2.85, 0.0], [0.25, 2.8, 0.0], [0.2, 2.8, 0.0], [0.2, 2.75, 0.0], [0.15, 2.7, 0.0], [0.3, 2.9, 0.0], [0.1, 2.7, 0.0]):
        delta = torch.tensor(cand)
        with torch.no_grad():
            model.base_model.class_embed[5].bias.copy_(bias0 + delta)
        tok.seq_len = 40
        with torch.no_grad():
            pred = model.forward_inference(samples=b["query_images"], support_coords=b["support_coords"],
                                           support_mask=b["support_masks"], skeleton_edges=b["support_skeletons"])
        tok.seq_len = 200
        kinds = set(pred["sequences"].reshape(-1).tolist())
        print("candidate", cand, "steps", pred["logits"].shape[1], "kinds", kinds)
        if pred["logits"].shape[1] >= 32 and len(kinds) == 3:
            break
    else:
        raise SystemExit("no candidate bias produced a mixed >= 32-step stream")
    npz("cfg5_512_decode.npz", logits=pred["logits"], coordinates=pred["coordinates"], sequences=pred["sequences"],
        bias_delta=delta, support_coords=b["support_coords"], support_masks=b["support_masks"])


# --------------------------------------------------------------------------------------------------------------
def crafted_predictions(rng, targets, plans, T):
    """Per sample: `plan` = (n_pred, sep_at, noise_px/512).  Coordinates of the first n_pred predicted <coord> tokens are
    the ground-truth keypoints plus noise (positions past the GT count get random points); one optional <sep> in the
    stream; <eos> after; logits one-hot-ish with distinct margins."""
    N = len(plans)
    logits = np.zeros((N, T, 3), dtype=np.float32)
    coords = rng.random((N, T, 2)).astype(np.float32)
    for i, (n_pred, sep_at, noise) in enumerate(plans):
        lab = targets["token_labels"][i].numpy()
        gt = targets["target_seq"][i].numpy()[lab == 0]
        t, k = 0, 0
        while t < T:
            if k < n_pred and t == sep_at:
                logits[i, t] = [0.1, 1.5, 0.2]; t += 1
                continue
            if k < n_pred:
                logits[i, t] = [2.0, 0.3, 0.1]
                if k < len(gt):
                    coords[i, t] = gt[k] + rng.normal(0, noise, 2)
                k += 1
            else:
                logits[i, t] = [0.2, 0.1, 1.7]
            t += 1
    return torch.from_numpy(logits), torch.from_numpy(np.clip(coords, 0, 1))


def eval_glue():
    args, tok, model, crit = refshim.build_reference()
    cfg = cape_ref.Cfg()
    from datasets.episodic_sampler import episodic_collate_fn as ref_collate
    from models.engine_cape import evaluate_cape
    rng = np.random.Generator(np.random.PCG64(77))
    tk = ref_tokenizer(tok)
    eps = [synth.make_episode(50 + i, 64, P, 2, 1, cfg, tokenizer=tk, category_id=c)
           for i, (P, c) in enumerate(((5, 3), (9, 7), (17, 3), (12, 9)))]
    batches = [ref_collate(eps[:3]), ref_collate(eps[3:]), ref_collate(eps[1:2])]
    batches[2] = {k: v for k, v in batches[2].items() if k != "query_metadata"}          # fallback: 512x512 boxes, no trim
    # plans per sample: (predicted keypoints, position of a <sep> or -1, noise sigma in normalised units)
    plans = [[(5, -1, 0.01), (3, -1, 0.05), (9, 2, 0.02), (14, -1, 0.2), (17, -1, 0.03), (10, 4, 0.08)],
             [(12, -1, 0.04), (12, 0, 0.3)],
             [(9, -1, 0.02), (9, -1, 0.1)]]
    Ts = [19, 200, 12]                                                                   # T < L, T = L, T < L
    preds = []
    for b, pl, T in zip(batches, plans, Ts):
        lg, co = crafted_predictions(rng, b["query_targets"], pl, T)
        preds.append({"logits": lg, "coordinates": co, "sequences": lg.argmax(-1)})

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = 0

        def forward_inference(self, samples, support_coords, support_mask, skeleton_edges=None):
            p = preds[self.calls]
            self.calls += 1
            return p

    stats = evaluate_cape(Fake(), crit, batches, torch.device("cpu"), compute_pck=True, pck_threshold=0.2)
    stats_nocrit = evaluate_cape(Fake(), None, batches, torch.device("cpu"), compute_pck=True, pck_threshold=0.2)
    per_batch = []
    for i in range(len(batches)):                                                        # counters batch by batch
        f = Fake(); f.calls = i
        s = evaluate_cape(f, None, batches[i:i + 1], torch.device("cpu"))
        per_batch.append([s["pck_num_correct"], s["pck_num_visible"]])
    arrs = {}
    for i, (b, p) in enumerate(zip(batches, preds)):
        arrs[f"b{i}_logits"], arrs[f"b{i}_coordinates"] = p["logits"], p["coordinates"]
        arrs[f"b{i}_support_coords"], arrs[f"b{i}_support_masks"] = b["support_coords"], b["support_masks"]
        arrs[f"b{i}_category_ids"] = b["category_ids"]
        for k, v in b["query_targets"].items():
            arrs[f"b{i}_t_{k}"] = v
    npz("eval_glue.npz", **arrs)
    meta = {"n_batches": len(batches),
            "query_metadata": [[{"bbox_width": m["bbox_width"], "bbox_height": m["bbox_height"], "visibility": list(m["visibility"])}
                                for m in b["query_metadata"]] if "query_metadata" in b else None for b in batches],
            "stats": {k: float(v) for k, v in stats.items()}, "stats_no_criterion": {k: float(v) for k, v in stats_nocrit.items()},
            "per_batch_correct_visible": per_batch}
    with open(os.path.join(OUT, "eval_glue.json"), "w") as f:
        json.dump(meta, f)
    print("eval stats", meta["stats"], per_batch)


# --------------------------------------------------------------------------------------------------------------
def train_loop():
    args, tok, model, crit = refshim.build_reference(["--dropout", "0"])
    cfg = cape_ref.Cfg(dropout=0.0)
    load_procedural(model)
    zero_dropout(model)
    from datasets.episodic_sampler import episodic_collate_fn as ref_collate
    from models.engine_cape import train_one_epoch_episodic
    tk = ref_tokenizer(tok)
    batches = [ref_collate([synth.make_episode(60 + i, 64, 9, 2, 1, cfg, tokenizer=tk, category_id=1 + i, n_invisible=2 * (i % 2))])
               for i in range(3)]
    param_dicts = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                   {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": args.lr_backbone}]
    opt = torch.optim.AdamW(param_dicts, lr=args.lr, weight_decay=args.weight_decay)
    before = {n: p.detach().clone() for n, p in model.named_parameters(remove_duplicate=False)}
    stats = train_one_epoch_episodic(model, crit, batches, opt, torch.device("cpu"), epoch=0, max_norm=args.clip_max_norm,
                                     print_freq=10, accumulation_steps=2)
    before = {n: before.get(n, None) for n in before}
    after = dict(model.named_parameters(remove_duplicate=False))
    dn = {n: float((after[n].detach() - before[n]).norm()) for n in before if after[n].requires_grad}
    picks = ["base_model.transformer.encoder.layers.0.linear1.weight", "base_model.transformer.decoder.layers.5.linear2.weight",
             "base_model.transformer.decoder.pos_trans.weight", "base_model.class_embed.5.weight",
             "base_model.backbone.0.body.layer4.2.conv3.weight", "base_model.input_proj.3.0.weight",
             "support_encoder.transformer_encoder.layers.2.linear1.weight", "support_encoder.gcn_layers.1.conv.weight",
             "base_model.transformer.decoder.layers.2.cross_attn.sampling_offsets.bias", "base_model.query_embed.weight"]
    heads = {"delta:" + n: (after[n].detach() - before[n]).reshape(-1)[:512] for n in picks}
    npz("train_loop.npz", dnorm_keys=jbytes(sorted(dn)), dnorm_vals=np.array([dn[k] for k in sorted(dn)]),
        stat_keys=jbytes(sorted(stats)), stat_vals=np.array([float(stats[k]) for k in sorted(stats)]),
        lr=np.array([args.lr, args.lr_backbone, args.weight_decay, args.clip_max_norm]), **heads)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["cfg4", "cfg5", "eval", "train"]
    for w in which:
        {"cfg4": cfg4, "cfg5": cfg5, "eval": eval_glue, "train": train_loop}[w]()
