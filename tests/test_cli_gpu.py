"""The training CLI end to end on synthetic episodes (reference `models/train_cape_episodic.py` flags): two epochs with
validation (cached AR decode + PCK), checkpoint naming / rotation, then resume for a third epoch."""
import argparse
import glob
import math
import os

import pytest

pytestmark = pytest.mark.gpu


def test_train_cli_two_epochs_and_resume(tmp_path):
    import cape_amd  # noqa: F401
    from cape_amd.models.train_cape_episodic import get_args_parser, main
    base = ["--use_geometric_encoder", "--use_gcn_preenc", "--dataset_name", "synthetic", "--image_size", "64", "--batch_size", "2",
            "--episodes_per_epoch", "4", "--val_episodes_per_epoch", "2", "--num_workers", "0", "--output_dir", str(tmp_path),
            "--print_freq", "0", "--accumulation_steps", "2"]
    parse = lambda extra: argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(base + extra)
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
    main(parse(["--epochs", "2"]))
    cks = sorted(glob.glob(str(tmp_path / "checkpoint_e*.pth")))
    assert [os.path.basename(c) for c in cks] == ["checkpoint_e000_lr1e-04_bs2_acc2_qpe2.pth", "checkpoint_e001_lr1e-04_bs2_acc2_qpe2.pth"]
    import torch
    from cape_amd.util.checkpoint import load_checkpoint
    ck = load_checkpoint(cks[-1])                                            # weights_only reader (nothing from the file is executed)
    assert "hip_rng_state" in ck and ck["hip_rng_state"].shape == (2,) and int(ck["hip_rng_state"][1]) > 0
    assert ck["epoch"] == 1 and math.isfinite(ck["train_stats"]["loss"]) and 0.0 <= ck["val_stats"]["pck"] <= 1.0
    assert len(ck["model"]) == 751
    main(parse(["--epochs", "3", "--resume", cks[-1]]))
    assert os.path.exists(tmp_path / "checkpoint_e002_lr1e-04_bs2_acc2_qpe2.pth")
    # f1: the checkpoint-evaluation script on the file the CLI just wrote (weights-only load -> model -> evaluate_cape -> metrics.json)
    from cape_amd.scripts import eval_cape_checkpoint
    m = eval_cape_checkpoint.main(["--checkpoint", cks[-1], "--num-episodes", "3", "--output-dir", str(tmp_path / "eval"), "--sort-by-pck", "id"])
    import json
    on_disk = json.load(open(tmp_path / "eval" / "metrics.json"))
    assert on_disk["pck_overall"] == m["pck_overall"] and 0.0 <= m["pck_overall"] <= 1.0 and m["total_visible"] > 0
    assert on_disk["epoch"] == 1 and on_disk["num_episodes"] == 3 and set(on_disk["pck_per_category"]) <= {str(i) for i in range(1, 11)}
    with pytest.raises(FileNotFoundError):
        eval_cape_checkpoint.main(["--checkpoint", str(tmp_path / "nope.pth")])


def test_validation_graph_capture_with_loader_workers(tmp_path):
    """decode-step hipGraphs are captured inside evaluate_cape while the validation DataLoader's worker /
    pin-memory threads are alive (CLI default num_workers = 2): the capture runs in thread_local error mode."""
    import cape_amd  # noqa: F401
    from cape_amd.models.train_cape_episodic import get_args_parser, main
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--dataset_name", "synthetic", "--image_size", "64", "--batch_size", "2",
         "--episodes_per_epoch", "2", "--val_episodes_per_epoch", "4", "--num_workers", "2", "--output_dir", str(tmp_path),
         "--print_freq", "0", "--epochs", "1"])
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
    hist = main(args)
    assert len(hist) == 1 and 0.0 <= hist[0]["val"]["pck"] <= 1.0 and math.isfinite(hist[0]["train"]["loss"])


def test_train_cli_on_mp100_files(tmp_path):
    """f2: the CLI on MP-100-style files (COCO annotations + images + category_splits.json): file loader -> episodic
    sampler -> collate -> training step at 512x512 (S = 5440 tokens) -> validation on the unseen categories."""
    import shutil
    import cape_amd  # noqa: F401
    from cape_amd.models.train_cape_episodic import get_args_parser, main
    from tests.test_data_path_cpu import make_dataset
    root = tmp_path / "mp100"
    root.mkdir()
    ann = make_dataset(root, n_per_cat=4)
    shutil.copy(ann, root / "annotations" / "mp100_split1_val.json")
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--dataset_name", "mp100", "--dataset_root", str(root), "--batch_size", "1",
         "--episodes_per_epoch", "2", "--val_episodes_per_epoch", "1", "--num_workers", "0", "--output_dir", str(tmp_path / "out"),
         "--print_freq", "0", "--epochs", "1", "--fixed_val_episodes"])
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
    hist = main(args)
    assert len(hist) == 1 and math.isfinite(hist[0]["train"]["loss"]) and 0.0 <= hist[0]["val"]["pck"] <= 1.0
    assert hist[0]["val"]["pck_num_visible"] > 0


def test_reference_format_checkpoint_through_the_eval_script(tmp_path, golden_dir, proc_sd):
    """Row f1: the reference-written checkpoint (tests/golden/ref_checkpoint.pth: pickled args, torch AdamW / SequentialLR
    state, RNG states, contaminated decoder keys; oracle/make_golden_r3.py) completed with the procedural tensors it leaves
    out, through `scripts/eval_cape_checkpoint.load_checkpoint_and_model` (weights-only read -> model from the stored args):
    the cached decode of that model reproduces the reference's e2e64_decode.npz stream, and the optimizer state loads into
    ArenaAdamW in the reference's parameter enumeration."""
    import json
    import numpy as np
    import torch
    import cape_amd  # noqa: F401
    from cape_amd.scripts import eval_cape_checkpoint
    from cape_amd.util.checkpoint import load_checkpoint
    from oracle import cape_ref, synth
    ck = load_checkpoint(os.path.join(golden_dir, "ref_checkpoint.pth"))
    meta = json.load(open(os.path.join(golden_dir, "ref_checkpoint.json")))
    d = np.load(os.path.join(golden_dir, "e2e64_decode.npz"))
    full = dict(proc_sd)
    full.update(ck["model"])                                  # the reference's tensors (and its contaminated keys) win
    for k in meta["stepped_keys"]:                            # the golden decode was made with the un-stepped procedural weights
        full[k] = proc_sd[k].clone()
    key, alias = "base_model.class_embed.5.bias", "base_model.transformer.decoder.class_embed.5.bias"
    full[key] = full[key] + torch.from_numpy(d["bias_delta"])
    full[alias] = full[key]
    ck["model"] = full
    path = tmp_path / "checkpoint_e001_ref.pth"
    torch.save(ck, path)
    model, args, tok, ck2 = eval_cape_checkpoint.load_checkpoint_and_model(str(path), torch.device("cuda:0"))
    assert ck2["epoch"] == 1 and args.scheduler == "cosine_warmrestarts" and len(ck2["model"]) == 751 + len(meta["contaminated_keys"])
    tok.seq_len = 40
    model.base_model.tokenizer.seq_len = 40
    cfg = cape_ref.Cfg()
    b = synth.make_batch(11, 2, 2, 64, 9, cfg, n_invisible=(2, 0))
    with torch.no_grad():
        p = model.forward_inference(samples=b["images"].cuda(), support_coords=b["support_coords"].cuda(),
                                    support_mask=b["support_mask"].cuda(), skeleton_edges=b["skeleton"])
    ref_logits, ref_seq = torch.from_numpy(d["logits"]), torch.from_numpy(d["sequences"]).long()
    assert p["logits"].shape == ref_logits.shape
    assert (p["logits"][:, :4].cpu() - ref_logits[:, :4]).abs().max() < 1e-3
    top2 = ref_logits.sort(-1).values
    clear = (top2[..., 2] - top2[..., 1]) > 5e-2
    assert torch.equal(p["sequences"].cpu()[clear], ref_seq[clear])
    # optimizer state in the reference's enumeration -> the arena optimizer
    from cape_amd.runtime.optimizer import ArenaAdamW
    model.train()
    opt = ArenaAdamW(model, lr=args.lr, lr_backbone=args.lr_backbone, weight_decay=args.weight_decay, max_norm=args.clip_max_norm)
    opt.load_state_dict(ck2["optimizer"])
    back = opt.state_dict()
    for i in meta["optimizer_state_indices"]:
        assert torch.allclose(back["state"][i]["exp_avg"].cpu(), ck2["optimizer"]["state"][i]["exp_avg"], atol=0), i
        assert float(back["state"][i]["exp_avg"].abs().max()) > 0


def test_straight_two_epochs_equal_one_plus_resume_plus_one(tmp_path, monkeypatch):
    """Row f3: 2 epochs in one run == 1 epoch, checkpoint, `--resume`, 1 more epoch (model weights; dropout on: the device dropout
    counter travels in the checkpoint).  Forward k-splits off (CAPE_DETERMINISTIC: two executions of one computation are
    compared); the remaining differences are arrival-order rounding of the backward's atomics."""
    import torch
    import cape_amd  # noqa: F401
    from cape_amd.hip import functional as HF
    from cape_amd.models.train_cape_episodic import get_args_parser, main
    from cape_amd.util.checkpoint import load_checkpoint
    monkeypatch.setattr(HF, "_DETERMINISTIC", True)
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"

    def run(out, extra):
        base = ["--use_geometric_encoder", "--use_gcn_preenc", "--dataset_name", "synthetic", "--image_size", "64", "--batch_size", "2",
                "--episodes_per_epoch", "4", "--val_episodes_per_epoch", "1", "--num_workers", "0", "--output_dir", str(out),
                "--print_freq", "0", "--seed", "11"]
        main(argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(base + extra))

    run(tmp_path / "a", ["--epochs", "2"])
    run(tmp_path / "b", ["--epochs", "1"])
    first = glob.glob(str(tmp_path / "b" / "checkpoint_e000_*.pth"))[0]
    run(tmp_path / "b", ["--epochs", "2", "--resume", first])
    a = load_checkpoint(glob.glob(str(tmp_path / "a" / "checkpoint_e001_*.pth"))[0])
    b = load_checkpoint(glob.glob(str(tmp_path / "b" / "checkpoint_e001_*.pth"))[0])
    a0 = load_checkpoint(glob.glob(str(tmp_path / "a" / "checkpoint_e000_*.pth"))[0])
    assert torch.equal(a["hip_rng_state"], b["hip_rng_state"])
    num = den = 0.0
    n_all = n_close = 0
    for k, v in a["model"].items():
        if not v.dtype.is_floating_point or not bool(torch.isfinite(v).all()):     # (the causal `attention_mask` buffer holds -inf)
            continue
        diff = (v.double() - b["model"][k].double())
        num += float((diff ** 2).sum()); den += float(((v.double() - a0["model"][k].double()) ** 2).sum())
        n_all += v.numel(); n_close += int((diff.abs() <= 1e-7).sum())
    assert den > 0 and (num / den) ** 0.5 < 2e-3, (num, den)           # distance between the two runs << the second epoch's own update
    assert n_close / n_all > 0.995, n_close / n_all
