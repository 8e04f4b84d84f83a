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
