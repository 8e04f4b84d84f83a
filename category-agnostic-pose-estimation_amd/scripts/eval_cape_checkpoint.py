#!/usr/bin/env python3
"""Standalone evaluation of a CAPE checkpoint on MI355X (surface of the reference's
`scripts/eval_cape_checkpoint.py`: flags :89-148, `load_checkpoint_and_model` :151-255, evaluation + metrics JSON :329-...).

    python -m cape_amd.scripts.eval_cape_checkpoint --checkpoint outputs/cape_run/checkpoint_e010_....pth --output-dir outputs/cape_eval

What it does: reads the checkpoint with a loader that executes nothing from the file (`util/checkpoint.py`:
`torch.load(weights_only=True)` + an allow-list for `argparse.Namespace` and numpy RNG arrays), rebuilds the model from the
stored training `args` (the tokenizer from `vocab_size` / `seq_len`, as `mp100_cape.py:118-121` does), loads the 751-key
`state_dict` (the keys of the reference; the "contaminated" `support_cross_attn_layers.*` / `support_attn_norms.*` keys of old
reference checkpoints are reported and ignored like the reference does), runs `evaluate_cape` (KV-cached autoregressive decode,
PCK@bbox) and writes `metrics.json`.  The visualisation flags of the reference are accepted and ignored: drawing needs the
image files on disk and matplotlib / cv2, which this package does not depend on.  Episodes come from the dataset the
checkpoint was trained on: MP-100 files (`datasets/mp100_cape.py`, `--dataset-root` to relocate) or `--dataset_name synthetic`.
"""
import argparse
import json
import math
import sys
from pathlib import Path

import numpy as np
import torch

CONTAMINATED = ("support_cross_attn_layers", "support_attn_norms")


def get_args_parser():
    p = argparse.ArgumentParser(description="Evaluate CAPE model checkpoint", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--checkpoint", required=True, type=str, help="Path to checkpoint (.pth file)")
    p.add_argument("--split", default="val", choices=["train", "val", "test"], help="Which split to evaluate on")
    p.add_argument("--dataset-root", default=None, type=str, help="Dataset root (if different from checkpoint args)")
    p.add_argument("--num-episodes", default=None, type=int, help="Number of episodes to evaluate (None = use default per split)")
    p.add_argument("--full-split", action="store_true", help="Evaluate on ALL images in the split (overrides --num-episodes)")
    p.add_argument("--eval_seed", default=123, type=int, help="Random seed for reproducible evaluation")
    p.add_argument("--num-queries-per-episode", default=None, type=int, help="Queries per episode (None = use checkpoint default)")
    p.add_argument("--pck-threshold", default=0.2, type=float, help="PCK threshold (fraction of bbox diagonal)")
    p.add_argument("--num-visualizations", default=50, type=int, help="(accepted, ignored: no image files on this path)")
    p.add_argument("--min-vis-per-category", default=10, type=int, help="(accepted, ignored)")
    p.add_argument("--visualize-top-pck", action="store_true", help="(accepted, ignored)")
    p.add_argument("--draw-skeleton", action="store_true", help="(accepted, ignored)")
    p.add_argument("--save-all-queries", action="store_true", help="(accepted, ignored)")
    p.add_argument("--organize-by-category", action="store_true", default=True, help="(accepted, ignored)")
    p.add_argument("--output-dir", default="outputs/cape_eval", type=str, help="Directory to save metrics")
    p.add_argument("--show-per-category", action="store_true", default=True, help="Show per-category PCK breakdown")
    p.add_argument("--sort-by-pck", choices=["asc", "desc", "id"], default="desc", help="Order of the per-category table")
    p.add_argument("--device", default=None, type=str, help="Device (cuda[:i]; default: cuda:0 -- there is no CPU path)")
    p.add_argument("--num-workers", default=0, type=int, help="Number of dataloader workers")
    return p


def load_checkpoint_and_model(checkpoint_path, device):
    """Checkpoint -> (model in eval mode on `device`, training args, tokenizer, checkpoint dict)."""
    from ..datasets import DiscreteTokenizerV2
    from ..models import build_model
    from ..models.cape_model import build_cape_model
    from ..util.checkpoint import load_checkpoint
    path = Path(checkpoint_path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    ck = load_checkpoint(str(path))
    if "args" not in ck or "model" not in ck:
        raise KeyError("checkpoint must hold 'args' (training argparse.Namespace) and 'model' (state_dict)")
    args = ck["args"]
    print(f"Checkpoint: {path.name}  epoch {ck.get('epoch', '?')}  best PCK {ck.get('best_pck', 'N/A')}")
    # the reference builds a dataset only to get its tokenizer (mp100_cape.py:118-121): same object from the stored args
    tok = DiscreteTokenizerV2(num_bins=int(math.sqrt(args.vocab_size)), seq_len=args.seq_len, add_cls=False)
    built = build_model(args, train=False, tokenizer=tok)
    base = built[0] if isinstance(built, tuple) else built
    model = build_cape_model(args, base)
    missing, unexpected = model.load_state_dict(ck["model"], strict=False)
    if missing:
        print(f"  Missing keys: {len(missing)}" + "".join(f"\n     - {k}" for k in missing[:5]))
    if unexpected:
        bad = [k for k in unexpected if any(c in k for c in CONTAMINATED)]
        print(f"  Unexpected keys: {len(unexpected)} ({len(bad)} from the reference's old state_dict contamination bug: ignored)")
    model.to(device).eval()
    print(f"Model loaded: {sum(p.numel() for p in model.parameters()) / 1e6:.1f}M parameters on {device}, "
          f"forward_inference: {hasattr(model, 'forward_inference')}")
    return model, args, tok, ck


def build_dataloader(args, tok, split, num_workers, num_episodes=None, num_queries=None, eval_seed=123):
    from ..datasets import episodic_collate_fn
    from ..datasets.synthetic import SyntheticEpisodes
    if getattr(args, "dataset_name", "synthetic") != "synthetic":
        # MP-100 files: the checkpoint's dataset_root (or --dataset-root) holds the annotations, images and category_splits.json
        from ..datasets import EpisodicDataset, build_mp100_cape
        K = num_queries if num_queries is not None else args.num_queries_per_episode
        n = num_episodes if num_episodes is not None else args.val_episodes_per_epoch
        ds = EpisodicDataset(build_mp100_cape(split, args), str(Path(args.dataset_root) / args.category_split_file), split=split,
                             num_queries_per_episode=K, episodes_per_epoch=n, seed=eval_seed, fixed_episodes=True,
                             load_support_images=False)
        return torch.utils.data.DataLoader(ds, 1, shuffle=False, collate_fn=episodic_collate_fn, num_workers=num_workers, pin_memory=True)
    n = num_episodes if num_episodes is not None else {"train": args.episodes_per_epoch, "val": args.val_episodes_per_epoch,
                                                       "test": args.val_episodes_per_epoch}[split]
    K = num_queries if num_queries is not None else args.num_queries_per_episode
    res = 512 if args.image_size == 512 else args.image_size
    ds = SyntheticEpisodes(tok, n, res, 17, K, seed=eval_seed + {"train": 0, "val": 999, "test": 1999}[split])
    return torch.utils.data.DataLoader(ds, 1, shuffle=False, collate_fn=episodic_collate_fn, num_workers=num_workers, pin_memory=True)


def per_category_table(loader, model, device, threshold):
    """One evaluation pass -> (stats, {category id: PCK}); the per-category numbers come from the PCKEvaluator that
    `evaluate_cape` itself fills (reference scripts/eval_cape_checkpoint.py:329-420 reads them from its evaluator the same way)."""
    from ..models import engine_cape
    return engine_cape.evaluate_cape(model, None, loader, device, compute_pck=True, pck_threshold=threshold, return_per_category=True)


def main(argv=None):
    a = get_args_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("cape_amd evaluates on MI355X only: no GPU visible (there is no CPU fallback)")
    device = torch.device(a.device if a.device else "cuda:0")
    if device.type != "cuda":
        raise RuntimeError(f"--device {a.device}: the HIP kernels need a GPU")
    torch.manual_seed(a.eval_seed); np.random.seed(a.eval_seed)
    model, args, tok, ck = load_checkpoint_and_model(a.checkpoint, device)
    if a.dataset_root:
        args.dataset_root = a.dataset_root
    loader = build_dataloader(args, tok, a.split, a.num_workers, None if a.full_split else a.num_episodes,
                              a.num_queries_per_episode, a.eval_seed)
    stats, per_cat = per_category_table(loader, model, device, a.pck_threshold)
    order = {"asc": lambda kv: kv[1], "desc": lambda kv: -kv[1], "id": lambda kv: kv[0]}[a.sort_by_pck]
    rows = sorted(per_cat.items(), key=order)
    print(f"PCK@{a.pck_threshold}: {stats['pck']:.4f} ({int(stats['pck_num_correct'])}/{int(stats['pck_num_visible'])} keypoints), "
          f"mean over categories {stats['pck_mean_categories']:.4f}")
    if a.show_per_category:
        for cid, v in rows:
            print(f"  category {cid:4d}: {v:.4f}")
    out = Path(a.output_dir)
    out.mkdir(parents=True, exist_ok=True)
    metrics = {"checkpoint": str(a.checkpoint), "epoch": ck.get("epoch"), "split": a.split, "pck_threshold": a.pck_threshold,
               "pck_overall": stats["pck"], "pck_mean_categories": stats["pck_mean_categories"],
               "total_correct": int(stats["pck_num_correct"]), "total_visible": int(stats["pck_num_visible"]),
               "pck_per_category": {str(k): v for k, v in per_cat.items()}, "num_episodes": len(loader), "eval_seed": a.eval_seed}
    with open(out / "metrics.json", "w") as f:
        json.dump(metrics, f, indent=2)
    print(f"metrics -> {out / 'metrics.json'}")
    return metrics


if __name__ == "__main__":
    main(sys.argv[1:])
