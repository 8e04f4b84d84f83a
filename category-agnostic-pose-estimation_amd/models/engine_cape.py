"""Training / evaluation engine of the CAPE episodic path on MI355X (interface of the reference's
`models/engine_cape.py`: `train_one_epoch_episodic` :48-301, `extract_keypoints_from_sequence` :304-391,
`evaluate_cape` :394-870) plus the one-process-per-GPU data-parallel loop the north star asks for
(`run_training`; the reference is single-process).

Per optimizer step the device work is: forward + fused criterion + backward kernels, bucketed RCCL
all-reduce of the flat gradient arena overlapped with backward (boundary micro-batch only), one
sum-of-squares kernel and one AdamW kernel per group.  The only host sync per micro-batch is the finite-loss
check the reference also performs (`engine_cape.py:204-209`)."""
import math
import os
import random
import sys
import time
from pathlib import Path
from typing import Iterable, Optional

import numpy as np
import torch

from ..util import misc as utils
from ..util.checkpoint import load_checkpoint, rng_restore, rng_snapshot
from ..util.eval_utils import PCKEvaluator
from ..hip import functional as HF

DEBUG_CAPE = os.environ.get("DEBUG_CAPE", "0") == "1"


_PIPELINES = {}            # (device, out_size) -> datasets.transforms.DeviceImagePipeline


def _query_images(batch, device):
    """The (N, 3, S, S) query batch on `device`: the collated images, or -- when the dataset defers its pixels
    (MP100CAPE(defer_pixels=True), the default of the MP-100 path) -- made on the GPU from the raw uint8 crops and their plans
    by two launches of csrc/augment.hip on the pipeline's stream (datasets/transforms.DeviceImagePipeline)."""
    if batch.get("query_images") is not None:
        return batch["query_images"].to(device, non_blocking=True)
    raw = batch["query_raw"]
    from ..datasets.transforms import DeviceImagePipeline
    size = raw[0][1].out_size
    key = (str(device), size)
    if key not in _PIPELINES:
        _PIPELINES[key] = DeviceImagePipeline(device, out_size=size)
    return _PIPELINES[key]([c for c, _ in raw], [p for _, p in raw])


def _to_device(batch, device):
    q = {k: v.to(device, non_blocking=True) for k, v in batch["query_targets"].items()}
    return (batch["support_coords"].to(device, non_blocking=True), batch["support_masks"].to(device, non_blocking=True),
            _query_images(batch, device), batch.get("support_skeletons", None), q)


def _scaled_dicts(loss_dict, weight_dict):
    plain = {k: v for k, v in loss_dict.items() if not k.startswith("_")}
    red = utils.reduce_dict({k: torch.as_tensor(v, dtype=torch.float32, device=loss_dict["_total"].device).detach()
                             for k, v in plain.items()})
    scaled = {k: v * weight_dict[k] for k, v in red.items() if k in weight_dict}
    unscaled = {f"{k}_unscaled": v for k, v in red.items()}
    return red, scaled, unscaled


def train_one_epoch_episodic(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable,
                             optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, max_norm: float = 0,
                             print_freq: int = 10, accumulation_steps: int = 1, scaler=None, ddp=None):
    """One epoch of episodic training with gradient accumulation.  `optimizer` is an `ArenaAdamW`
    (clipping happens inside its fused step; `max_norm` is forwarded to it); `ddp` an optional
    `EpisodeDataParallel`.  `scaler` must be None: the MI355X path keeps fp32 storage and accumulation, with the GEMMs as a
    bf16x3 split by default (CAPE_GEMM_PRECISION=f32 selects exact fp32 MFMA); plain fp16/bf16 autocast misses the parity
    bar of 1e-3 on logits."""
    if scaler is not None:
        raise ValueError("a GradScaler has no role on the MI355X path: --use_amp selects bf16 MFMA products with fp32 accumulation "
                         "(run_training), which need no loss scaling")
    model.train()
    criterion.train()
    if hasattr(optimizer, "max_norm"):
        optimizer.max_norm = max_norm
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    optimizer.zero_grad()
    world_scale = ddp.loss_scale if ddp is not None else 1.0
    rng = HF.Runtime.get_rng(device)
    n_batches = 0
    pending = 0

    def do_step():
        if ddp is not None:
            ddp.finish()
        optimizer.step()
        optimizer.zero_grad()

    for batch_idx, batch in enumerate(data_loader):
        support_coords, support_masks, query_images, skeletons, targets = _to_device(batch, device)
        boundary = (batch_idx + 1) % accumulation_steps == 0
        rng.advance()
        ctx = ddp.no_sync() if (ddp is not None and not boundary) else _null()
        with ctx:
            outputs = model(samples=query_images, support_coords=support_coords, support_mask=support_masks,
                            targets=targets, skeleton_edges=skeletons)
            loss_dict = criterion(outputs, targets)
            losses = loss_dict["_total"]
            red, scaled, unscaled = _scaled_dicts(loss_dict, criterion.weight_dict)
            loss_value = float(sum(scaled.values()))            # the per-iteration sync of engine_cape.py:204
            if not math.isfinite(loss_value):
                print(f"Loss is {loss_value}, stopping training")
                print(red)
                sys.exit(1)
            (losses * (world_scale / accumulation_steps)).backward()
        pending += 1
        if boundary:
            do_step()
            pending = 0
        metric_logger.update(loss=loss_value, **scaled, **unscaled)
        metric_logger.update(lr=optimizer.param_groups[0]["lr"])
        n_batches = batch_idx + 1
        if print_freq and batch_idx % print_freq == 0 and utils.is_main_process():
            print(f"Epoch [{epoch}] it {batch_idx}: loss {loss_value:.4f} lr {optimizer.param_groups[0]['lr']:.6f}")
    if pending:
        # tail flush of engine_cape.py:280-295; the boundary exchange was skipped for these micro-batches
        if ddp is not None:
            ddp.finish()
        optimizer.step()
        optimizer.zero_grad()
    metric_logger.synchronize_between_processes()
    return {k: m.global_avg for k, m in metric_logger.meters.items()}


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def extract_keypoints_from_sequence(pred_coords, token_labels, mask, max_keypoints: Optional[int] = None):
    """Coordinates of the valid tokens whose label is <coord>, zero-padded to the batch maximum."""
    per = []
    for i in range(pred_coords.shape[0]):
        c, l = pred_coords[i][mask[i]], token_labels[i][mask[i]]
        k = c[l == 0]
        if max_keypoints is not None and len(k) > max_keypoints:
            k = k[:max_keypoints]
        per.append(k)
    n = max((len(k) for k in per), default=0)
    out = torch.zeros(len(per), n, 2, device=pred_coords.device)
    for i, k in enumerate(per):
        out[i, :len(k)] = k
    return out


def extract_keypoints_from_predictions(pred_coords, pred_logits, max_keypoints: Optional[int] = None):
    """`util/sequence_utils.py:8-65`: positions whose *predicted* token type is <coord>."""
    types = pred_logits.argmax(-1)
    per = []
    for i in range(pred_coords.shape[0]):
        k = pred_coords[i][types[i] == 0]
        if max_keypoints is not None and len(k) > max_keypoints:
            k = k[:max_keypoints]
        per.append(k)
    n = max((len(k) for k in per), default=0)
    out = torch.zeros(len(per), n, 2, device=pred_coords.device)
    for i, k in enumerate(per):
        out[i, :len(k)] = k
    return out


@torch.no_grad()
def evaluate_cape(model, criterion, data_loader, device, compute_pck=True, pck_threshold=0.2, return_per_category=False):
    """Autoregressive validation: KV-cached decode -> pad/trim to the target length -> validation loss;
    PCK@bbox on keypoints extracted by predicted token types (GT keypoints by GT labels), trimmed / zero
    padded to the category's keypoint count, scaled by 512 like the reference (`engine_cape.py:773-841`).
    `return_per_category=True` (not a reference argument; the checkpoint-evaluation script's table, reference
    scripts/eval_cape_checkpoint.py:329-420) returns (stats, {category id: PCK}) instead of stats."""
    model.eval()
    if criterion is not None:
        criterion.eval()
    metric_logger = utils.MetricLogger(delimiter="  ")
    pck = PCKEvaluator(threshold=pck_threshold) if compute_pck else None
    infer = model if hasattr(model, "forward_inference") else getattr(model, "module", None)
    if infer is None or not hasattr(infer, "forward_inference"):
        raise RuntimeError("Model does not have forward_inference method!")
    for batch in data_loader:
        support_coords, support_masks, query_images, skeletons, targets = _to_device(batch, device)
        pred = infer.forward_inference(samples=query_images, support_coords=support_coords, support_mask=support_masks,
                                       skeleton_edges=skeletons)
        logits, coords = pred["logits"], pred["coordinates"]
        B, T = logits.shape[:2]
        L = targets["target_seq"].shape[1]
        if T < L:
            logits_p = torch.cat([logits, torch.zeros(B, L - T, logits.shape[-1], device=device)], 1)
            coords_p = torch.cat([coords, torch.zeros(B, L - T, 2, device=device)], 1)
        else:
            logits_p, coords_p = logits[:, :L], coords[:, :L]
        if criterion is not None:
            ld = criterion({"pred_logits": logits_p.contiguous(), "pred_coords": coords_p.contiguous()}, targets)
            red, scaled, unscaled = _scaled_dicts(ld, criterion.weight_dict)
            metric_logger.update(loss=float(sum(scaled.values())), **scaled, **unscaled)
        if pck is not None:
            gt_k = extract_keypoints_from_sequence(targets["target_seq"], targets["token_labels"], targets["mask"])
            pr_k = extract_keypoints_from_predictions(coords, logits)
            meta = batch.get("query_metadata") or []
            if meta:
                bw, bh, vis_l, p_l, g_l = [], [], [], [], []
                for i, m in enumerate(meta):
                    bw.append(m.get("bbox_width", 512.0)); bh.append(m.get("bbox_height", 512.0))
                    vis = m.get("visibility", [])
                    n = len(vis)
                    p = pr_k[i, :n] if pr_k.shape[1] >= n else torch.cat([pr_k[i], torch.zeros(n - pr_k.shape[1], 2, device=device)])
                    p_l.append(p * 512.0); g_l.append(gt_k[i, :n] * 512.0); vis_l.append(vis)
                pck.add_batch(p_l, g_l, bw, bh, category_ids=batch.get("category_ids"), visibility=vis_l)
            else:
                pck.add_batch([k * 512.0 for k in pr_k], [k * 512.0 for k in gt_k], [512.0] * B, [512.0] * B,
                              category_ids=batch.get("category_ids"))
    metric_logger.synchronize_between_processes()
    stats = {k: m.global_avg for k, m in metric_logger.meters.items()}
    if pck is not None:
        pck.synchronize_between_processes()
        r = pck.get_results()
        stats.update(pck=r["pck_overall"], pck_mean_categories=r["mean_pck_categories"],
                     pck_num_correct=r["total_correct"], pck_num_visible=r["total_visible"])
    for k in ("loss", "loss_ce", "loss_coords"):
        stats.setdefault(k, 0.0)
    if return_per_category:
        return stats, (dict(r.get("pck_per_category", {})) if pck is not None else {})
    return stats


# ------------------------------------------------------------------------------------------------
# driver used by the CLI (models/train_cape_episodic.py)
# ------------------------------------------------------------------------------------------------
def cleanup_old_checkpoints(output_dir, pattern, keep_last_n=3, exclude_pattern=None):
    files = [p for p in Path(output_dir).glob(pattern) if not (exclude_pattern and exclude_pattern in p.name)]
    files.sort(key=lambda p: p.stat().st_mtime)
    gone = []
    for p in files[:-keep_last_n] if len(files) > keep_last_n else []:
        try:
            p.unlink(); gone.append(p)
        except OSError:
            pass
    return gone


def build_scheduler(optimizer, args, steps_per_epoch=None):
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts, LinearLR, MultiStepLR, OneCycleLR, SequentialLR
    if args.scheduler == "multistep":
        main = MultiStepLR(optimizer, [int(x) for x in args.lr_drop.split(",")])
    elif args.scheduler == "cosine_warmrestarts":
        main = CosineAnnealingWarmRestarts(optimizer, T_0=args.T_0, T_mult=args.T_mult, eta_min=args.eta_min)
    elif args.scheduler == "onecycle":
        return OneCycleLR(optimizer, max_lr=args.lr * 10, epochs=args.epochs, steps_per_epoch=steps_per_epoch, pct_start=0.1,
                          anneal_strategy="cos")
    else:
        raise ValueError(f"Unknown scheduler: {args.scheduler}")
    if args.warmup_epochs > 0:
        return SequentialLR(optimizer, [LinearLR(optimizer, start_factor=0.1, total_iters=args.warmup_epochs), main],
                            milestones=[args.warmup_epochs])
    return main


def init_distributed():
    """One process per GPU (torchrun env); backend nccl == RCCL over xGMI on ROCm, gloo on CPU hosts."""
    import torch.distributed as dist
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ or int(os.environ["WORLD_SIZE"]) < 2:
        return 0, 1, 0
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def run_training(args):
    from ..datasets import DiscreteTokenizerV2, episodic_collate_fn
    from ..datasets.synthetic import SyntheticEpisodes
    from ..runtime.data_parallel import EpisodeDataParallel
    from ..runtime.optimizer import ArenaAdamW
    from . import build_model
    from .cape_losses import build_cape_criterion
    from .cape_model import build_cape_model

    rank, world, local = init_distributed()
    if not torch.cuda.is_available():
        raise RuntimeError("cape_amd trains on MI355X only: no GPU visible (there is no CPU fallback)")
    device = torch.device(f"cuda:{local}")
    # host-bound step: the autograd engine stays on the calling thread (no worker hand-off per backward call)
    torch.autograd.set_multithreading_enabled(False)
    args.device = str(device)
    seed = args.seed + rank
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    HF.Runtime.seed(seed, device)
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    tok = DiscreteTokenizerV2(num_bins=int(math.sqrt(args.vocab_size)), seq_len=args.seq_len, add_cls=False)
    base, _ = build_model(args, tokenizer=tok)
    criterion = build_cape_criterion(args, num_classes=3).to(device)
    model = build_cape_model(args, base).to(device)
    if args.batch_size % world:
        raise ValueError(f"--batch_size {args.batch_size} (episodes per global batch) must be divisible by the world size {world}")
    per_rank = args.batch_size // world
    if args.dataset_name == "synthetic":
        res = 512 if args.image_size == 512 else args.image_size
        train_ds = SyntheticEpisodes(tok, args.episodes_per_epoch, res, 17, args.num_queries_per_episode, seed=args.seed)
        val_ds = SyntheticEpisodes(tok, args.val_episodes_per_epoch, res, 17, args.num_queries_per_episode, seed=args.val_seed + 999)
    else:
        # MP-100 files (reference train_cape_episodic.py:351-354, :417, :465-516): COCO-style annotations under --dataset_root,
        # episodes of one category each; validation on the unseen categories with batch size 1
        from ..datasets import EpisodicDataset, build_mp100_cape
        split_file = str(Path(args.dataset_root) / args.category_split_file)
        # pixels are made on the GPU (raw crops + plans travel through the loader); CAPE_HOST_AUGMENT=1 keeps them on the host cores
        defer = os.environ.get("CAPE_HOST_AUGMENT", "0") != "1" and not args.image_norm
        train_ds = EpisodicDataset(build_mp100_cape("train", args, defer_pixels=defer), split_file, split="train",
                                   num_queries_per_episode=args.num_queries_per_episode, episodes_per_epoch=args.episodes_per_epoch,
                                   seed=args.seed, load_support_images=False)
        fixed_val = getattr(args, "fixed_val_episodes", False)
        val_ds = EpisodicDataset(build_mp100_cape("val", args, defer_pixels=defer), split_file, split="val",
                                 num_queries_per_episode=args.num_queries_per_episode, episodes_per_epoch=args.val_episodes_per_epoch,
                                 seed=(args.val_seed if fixed_val else args.seed + 999), fixed_episodes=fixed_val,
                                 load_support_images=False)
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds, world, rank, shuffle=False) if world > 1 else None
    vsampler = torch.utils.data.distributed.DistributedSampler(val_ds, world, rank, shuffle=False) if world > 1 else None
    train_loader = torch.utils.data.DataLoader(train_ds, per_rank, sampler=sampler, collate_fn=episodic_collate_fn,
                                               num_workers=args.num_workers, drop_last=True, pin_memory=True)
    val_loader = torch.utils.data.DataLoader(val_ds, 1, sampler=vsampler, collate_fn=episodic_collate_fn,
                                             num_workers=args.num_workers, pin_memory=True)
    optimizer = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay,
                           max_norm=args.clip_max_norm)
    if getattr(args, "use_amp", False):
        # the reference's --use_amp = torch.cuda.amp.autocast + GradScaler (engine_cape.py:164-179).  Its MI355X counterpart: every GEMM
        # as ONE bf16 MFMA per product (fp32 master weights, fp32 accumulation, fp32 everything else; bf16 has fp32's exponent range,
        # so no loss scaling).  A throughput option, NOT the parity path: logits deviate ~2e-2 from the fp32 reference.
        from ..hip import ops as _ops
        _ops.set_gemm_precision("bf16")
        if utils.is_main_process():
            print("--use_amp: GEMMs as single bf16 MFMA products (fp32 accumulate); outside the 1e-3 parity tolerance by design")
    ddp = EpisodeDataParallel(model, optimizer) if world > 1 else None
    lr_scheduler = build_scheduler(optimizer, args, steps_per_epoch=len(train_loader))
    best_pck, no_improve = 0.0, 0
    if args.resume:
        ck = load_checkpoint(args.resume)           # weights_only loader (util/checkpoint.py): nothing in the file is executed
        model.load_state_dict(ck["model"], strict=False)
        if "optimizer" in ck:
            optimizer.load_state_dict(ck["optimizer"])
        if "lr_scheduler" in ck:
            lr_scheduler.load_state_dict(ck["lr_scheduler"])
        args.start_epoch = ck.get("epoch", -1) + 1
        best_pck, no_improve = ck.get("best_pck", 0.0), ck.get("epochs_without_improvement", 0)
        rng_restore(ck, HF.Runtime.get_rng(device))   # host generators, torch's device generators and the HIP dropout counter
    history = []
    for epoch in range(args.start_epoch, args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0 = time.time()
        train_stats = train_one_epoch_episodic(model, criterion, train_loader, optimizer, device, epoch, args.clip_max_norm,
                                               args.print_freq, args.accumulation_steps, None, ddp)
        torch.cuda.synchronize()
        dt = time.time() - t0
        lr_scheduler.step()
        val_stats = evaluate_cape(model, criterion, val_loader, device)
        eps = len(train_loader) * args.batch_size / dt
        if utils.is_main_process():
            print(f"epoch {epoch}: train loss {train_stats.get('loss', 0):.4f}  val PCK {val_stats.get('pck', 0):.4f}  "
                  f"{eps:.2f} episodes/s")
            name = (f"checkpoint_e{epoch:03d}_lr{args.lr:.0e}_bs{args.batch_size}_acc{args.accumulation_steps}_"
                    f"qpe{args.num_queries_per_episode}.pth")
            ck = {"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(),
                  "scaler": None, "epoch": epoch, "args": args, "train_stats": train_stats, "val_stats": val_stats,
                  "best_pck": best_pck, "epochs_without_improvement": no_improve, **rng_snapshot(HF.Runtime.get_rng(device))}
            torch.save(ck, Path(args.output_dir) / name)
            cleanup_old_checkpoints(args.output_dir, "checkpoint_e*.pth", 3, "best")
        pck_now, pck_mean = val_stats.get("pck", 0.0), val_stats.get("pck_mean_categories", 0.0)
        if pck_now > best_pck:
            best_pck, no_improve = pck_now, 0
            if utils.is_main_process():
                ck2 = dict(ck)
                ck2.update(val_pck=pck_now, val_pck_mean=pck_mean, best_pck=best_pck)
                torch.save(ck2, Path(args.output_dir) / f"checkpoint_best_pck_e{epoch:03d}_pck{pck_now:.4f}_meanpck{pck_mean:.4f}.pth")
        else:
            no_improve += 1
        history.append({"epoch": epoch, "train": train_stats, "val": val_stats, "episodes_per_s": eps})
        if args.early_stopping_patience > 0 and no_improve >= args.early_stopping_patience:
            break
    return history
