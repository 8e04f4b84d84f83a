"""TEST INFRASTRUCTURE ONLY -- round-3 golden vectors from the REAL reference (build container only;
/root/reference is imported in place through oracle/refshim.py, nothing is copied).  Fixtures are data: seeds /
inputs and the reference's outputs.  Re-run:  python -m oracle.make_golden_r3 [e2e256] [cfg3] [ckpt]

  e2e256_grads.npz      the headline shape's BACKWARD: the episode of e2e256.npz (256x256, 17 keypoints, N = 2 query images)
                        teacher-forced with autograd on -- total loss, the gradient norm of every trained tensor, 8 gradient
                        slices (trunk, input_proj, encoder FFN / MSDA projections, decoder self- / support- / deformable
                        attention, support encoder).
  cfg3_5shot_256.npz    BASELINE configs[2] as a whole: two 5-shot episodes (17 keypoints, 2 queries each, 256x256) through
                        the reference's own `episodic_collate_fn` (mean-pooled support, datasets/episodic_sampler.py:438-442),
                        GCN pre-encoder on -> teacher-forced step: 6-layer logits / coords (first 24 positions), 19 losses,
                        gradient norms and 6 gradient slices.  The collated support tensors are stored so that the
                        product's collate can be compared too.
  ref_checkpoint.pth    a checkpoint written EXACTLY as models/train_cape_episodic.py:863-890 writes it (`torch.save` of
  ref_checkpoint.json   {model, optimizer, lr_scheduler, epoch, args (pickled argparse.Namespace), rng states, best_pck, ...}) by the
                        reference's own objects, plus the contaminated `support_cross_attn_layers.*` keys that old
                        checkpoints carry (:640-660).  To stay small (< 200 KB) only a handful of `model` tensors are
                        stored (float16 would change values: they are stored as-is, fp32); the loader test fills the rest
                        procedurally -- the json lists which keys are real and their checksums.
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
from oracle.make_golden_r2 import jbytes, stack_layers  # noqa: E402

os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
warnings.filterwarnings("ignore")

E2E256_HEADS = ["base_model.backbone.0.body.layer2.3.conv2.weight", "base_model.backbone.0.body.layer4.0.downsample.0.weight",
                "base_model.input_proj.2.0.weight", "base_model.transformer.encoder.layers.3.linear1.weight",
                "base_model.transformer.encoder.layers.0.self_attn.value_proj.weight",
                "base_model.transformer.decoder.layers.2.self_attn.in_proj_weight",
                "base_model.transformer.decoder.layers.5.cross_attn.sampling_offsets.weight",
                "support_encoder.transformer_encoder.layers.1.self_attn.in_proj_weight"]
CFG3_HEADS = ["base_model.backbone.0.body.layer3.5.conv3.weight", "base_model.transformer.encoder.layers.5.linear2.weight",
              "base_model.transformer.decoder.layers.0.support_attn.in_proj_weight",
              "base_model.transformer.decoder.layers.3.attn_k.weight", "support_encoder.gcn_layers.0.conv.weight",
              "support_encoder.coord_mlp.0.weight"]


def step_with_grads(model, crit, images, support_coords, support_mask, targets, skeleton, heads):
    model.eval()
    model.zero_grad(set_to_none=True)
    out = model(samples=images, support_coords=support_coords, support_mask=support_mask, targets=targets, skeleton_edges=skeleton)
    ld = crit(out, targets)
    loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    loss.backward()
    named = dict(model.named_parameters(remove_duplicate=False))
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    hd = {"gradhead:" + n: named[n].grad.reshape(-1)[:256] for n in heads}
    return dict(logits=stack_layers(out, "pred_logits")[:, :, :24], coords=stack_layers(out, "pred_coords")[:, :, :24], loss=loss,
                loss_keys=jbytes(sorted(ld.keys())), loss_vals=np.array([float(ld[k]) for k in sorted(ld.keys())]),
                gnorm_keys=jbytes(sorted(gn)), gnorm_vals=np.array([gn[k] for k in sorted(gn)]), **hd)


def e2e256():
    args, tok, model, crit = refshim.build_reference()
    cfg = cape_ref.Cfg()
    load_procedural(model)
    batch = synth.make_batch(23, 1, 2, 256, 17, cfg, n_invisible=(2,), tokenizer=ref_tokenizer(tok))
    r = step_with_grads(model, crit, batch["images"], batch["support_coords"], batch["support_mask"], batch["targets"],
                        batch["skeleton"], E2E256_HEADS)
    old = np.load(os.path.join(OUT, "e2e256.npz"))                       # the forward-only fixture of round 1: same episode
    assert np.abs(old["logits"] - r["logits"].detach().numpy()).max() < 1e-5
    npz("e2e256_grads.npz", **r)


def cfg3():
    args, tok, model, crit = refshim.build_reference()
    cfg = cape_ref.Cfg()
    load_procedural(model)
    from datasets.episodic_sampler import episodic_collate_fn as ref_collate
    tk = ref_tokenizer(tok)
    eps = [synth.make_episode(80 + i, 256, 17, 2, 5, cfg, tokenizer=tk, category_id=2 + 3 * i, n_invisible=2 * i) for i in range(2)]
    b = ref_collate(eps)
    assert b["support_coords"].shape == (4, 17, 2) and b["query_images"].shape == (4, 3, 256, 256)
    r = step_with_grads(model, crit, b["query_images"], b["support_coords"], b["support_masks"], b["query_targets"],
                        b["support_skeletons"], CFG3_HEADS)
    npz("cfg3_5shot_256.npz", support_coords=b["support_coords"], support_masks=b["support_masks"], **r)


def ckpt():
    """The checkpoint dict of train_cape_episodic.py:863-890, produced by the reference's own objects."""
    import random
    args, tok, model, crit = refshim.build_reference(["--epochs", "3"])
    load_procedural(model)
    param_dicts = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                   {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": args.lr_backbone}]
    optimizer = torch.optim.AdamW(param_dicts, lr=args.lr, weight_decay=args.weight_decay)
    # build_scheduler of train_cape_episodic.py:561-605 with the parser's defaults (cosine_warmrestarts behind a 5-epoch warm-up)
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts, LinearLR, SequentialLR
    assert args.scheduler == "cosine_warmrestarts" and args.warmup_epochs > 0
    lr_scheduler = SequentialLR(optimizer, [LinearLR(optimizer, start_factor=0.1, total_iters=args.warmup_epochs),
                                            CosineAnnealingWarmRestarts(optimizer, T_0=args.T_0, T_mult=args.T_mult, eta_min=args.eta_min)],
                                milestones=[args.warmup_epochs])
    full = model.state_dict()
    # a handful of real tensors (small ones + slices are not possible: whole tensors only); the loader test rebuilds the others
    real = ["base_model.class_embed.5.bias", "base_model.class_embed.5.weight", "base_model.query_embed.weight",
            "base_model.transformer.level_embed", "base_model.transformer.decoder.pos_trans_norm.weight",
            "base_model.input_proj.0.1.weight", "support_encoder.coord_mlp.0.weight", "support_encoder.coord_mlp.0.bias",
            "base_model.transformer.decoder.layers.0.norm2.bias", "base_model.backbone.0.body.layer1.0.bn1.running_var"]
    # two parameters take one real AdamW step, so that `optimizer` carries state in torch's own layout (index -> exp_avg, ...)
    named = dict(model.named_parameters())                  # (deduplicated: the heads are registered under the decoder's names)
    for n in ("base_model.transformer.decoder.class_embed.5.bias", "base_model.query_embed.weight"):
        named[n].grad = torch.full_like(named[n], 0.01)
    optimizer.step()
    lr_scheduler.step()
    sd = {k: full[k].clone() for k in real}               # after the step: the file is self-consistent
    # contamination of old checkpoints (:640-660): temporary decoder attributes that were once swept into state_dict()
    sd["base_model.transformer.decoder.support_cross_attn_layers.0.in_proj_weight"] = torch.full((8, 8), 0.25)
    sd["base_model.transformer.decoder.support_cross_attn_layers.0.out_proj.bias"] = torch.arange(8, dtype=torch.float32)
    sd["base_model.transformer.decoder.support_attn_norms.1.weight"] = torch.ones(4)
    random.seed(5); np.random.seed(6); torch.manual_seed(7)
    checkpoint = {                                           # key for key train_cape_episodic.py:863-890 (CPU run: no cuda_rng_state)
        "model": sd,
        "optimizer": optimizer.state_dict(),
        "lr_scheduler": lr_scheduler.state_dict(),
        "scaler": None,
        "epoch": 1,
        "args": args,
        "train_stats": {"loss": 3.25, "loss_ce": 0.5, "loss_coords": 0.55, "lr": args.lr},
        "val_stats": {"loss": 3.5, "pck": 0.4321, "pck_mean_categories": 0.41},
        "best_pck": 0.4321,
        "epochs_without_improvement": 2,
        "rng_state": torch.get_rng_state(),
        "np_rng_state": np.random.get_state(),
        "py_rng_state": random.getstate(),
    }
    path = os.path.join(OUT, "ref_checkpoint.pth")
    torch.save(checkpoint, path)
    meta = {"real_keys": real, "checksums": {k: float(sd[k].double().sum()) for k in real},
            "stepped_keys": ["base_model.class_embed.5.bias", "base_model.query_embed.weight"],
            "contaminated_keys": [k for k in sd if "support_cross_attn" in k or "support_attn_norm" in k],
            "optimizer_state_indices": sorted(int(i) for i in checkpoint["optimizer"]["state"]),
            "args": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in sorted(vars(args).items())},
            "epoch": 1, "best_pck": 0.4321, "epochs_without_improvement": 2, "bytes": os.path.getsize(path)}
    with open(os.path.join(OUT, "ref_checkpoint.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("wrote ref_checkpoint.pth", meta["bytes"], "bytes")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["e2e256", "cfg3", "ckpt"]
    for w in which:
        {"e2e256": e2e256, "cfg3": cfg3, "ckpt": ckpt}[w]()
