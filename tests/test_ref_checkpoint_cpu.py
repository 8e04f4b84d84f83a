"""Row f1: a checkpoint in the REFERENCE's format -- written by the reference's own objects exactly as
models/train_cape_episodic.py:863-890 writes it (oracle/make_golden_r3.py ckpt: pickled argparse.Namespace `args`,
torch.optim.AdamW state, SequentialLR state, host RNG states, the contaminated decoder keys of :640-660) -- read by the
product's loader, which executes nothing from the file (util/checkpoint.py: torch.load(weights_only=True) + allow-list)."""
import json
import os
import random

import numpy as np
import torch

import cape_amd  # noqa: F401


def _load(golden_dir):
    from cape_amd.util.checkpoint import load_checkpoint
    ck = load_checkpoint(os.path.join(golden_dir, "ref_checkpoint.pth"))
    meta = json.load(open(os.path.join(golden_dir, "ref_checkpoint.json")))
    return ck, meta


def test_reference_checkpoint_reads_weights_only(golden_dir):
    ck, meta = _load(golden_dir)
    # key for key what the reference's save writes on a CPU host
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "scaler", "epoch", "args", "train_stats", "val_stats", "best_pck",
                       "epochs_without_improvement", "rng_state", "np_rng_state", "py_rng_state"}
    assert ck["epoch"] == meta["epoch"] and ck["best_pck"] == meta["best_pck"] and ck["scaler"] is None
    args = ck["args"]
    for k, v in meta["args"].items():                     # the pickled Namespace came through the allow-list intact
        got = getattr(args, k)
        assert got == v or str(got) == v, k
    for k in meta["real_keys"]:
        assert abs(float(ck["model"][k].double().sum()) - meta["checksums"][k]) < 1e-9, k
    assert sorted(int(i) for i in ck["optimizer"]["state"]) == meta["optimizer_state_indices"]
    assert len(ck["optimizer"]["param_groups"]) == 2 and ck["optimizer"]["param_groups"][1]["initial_lr"] == args.lr_backbone


def test_reference_checkpoint_model_and_parser_defaults(golden_dir, proc_sd):
    """The stored args rebuild the product model (same flags as the product parser's defaults), the real tensors load, the
    contaminated keys are the only unexpected ones, and the RNG states restore."""
    import argparse
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    from cape_amd.util.checkpoint import rng_restore
    ck, meta = _load(golden_dir)
    args = ck["args"]
    mine = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(["--use_geometric_encoder", "--use_gcn_preenc", "--device", "cpu", "--epochs", "3"])
    assert vars(mine).keys() == vars(args).keys()
    # (dataset_root's default is a path relative to each parser's own file)
    assert {k for k in vars(args) if getattr(args, k) != getattr(mine, k)} <= {"dataset_root"}
    tok = DiscreteTokenizerV2(int(args.vocab_size ** 0.5), args.seq_len, add_cls=False)
    base, _ = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base)
    missing, unexpected = model.load_state_dict(ck["model"], strict=False)
    assert sorted(unexpected) == sorted(meta["contaminated_keys"])
    assert len(missing) == len(model.state_dict()) - len(meta["real_keys"])
    sd = model.state_dict()
    for k in meta["real_keys"]:
        assert torch.equal(sd[k], ck["model"][k]), k
        if k not in meta["stepped_keys"]:                # the reference model carried the procedural weights (two took an AdamW step)
            assert torch.equal(sd[k], proc_sd[k]), k
        else:
            assert not torch.equal(sd[k], proc_sd[k]) and (sd[k] - proc_sd[k]).abs().max() < 2e-4, k
    rng_restore(ck)
    a = (torch.rand(3), np.random.rand(3), random.random())
    random.seed(5); np.random.seed(6); torch.manual_seed(7)
    b = (torch.rand(3), np.random.rand(3), random.random())
    assert torch.equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_reference_scheduler_state_loads_into_product_scheduler(golden_dir):
    """lr_scheduler state written by the reference's SequentialLR(LinearLR, CosineAnnealingWarmRestarts) loads into the one
    the product's build_scheduler makes from the same args, and the next epoch's learning rates agree."""
    from cape_amd.models.engine_cape import build_scheduler
    ck, _ = _load(golden_dir)
    args = ck["args"]
    ps = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))]
    opt = torch.optim.AdamW([{"params": [ps[0]]}, {"params": [ps[1]], "lr": args.lr_backbone}], lr=args.lr, weight_decay=args.weight_decay)
    sch = build_scheduler(opt, args, steps_per_epoch=4)
    sch.load_state_dict(ck["lr_scheduler"])
    assert sch.last_epoch == ck["lr_scheduler"]["last_epoch"] == 1
    assert [round(x, 12) for x in sch.get_last_lr()] == [round(x, 12) for x in ck["lr_scheduler"]["_last_lr"]]
