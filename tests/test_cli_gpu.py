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
    ck = torch.load(cks[-1], map_location="cpu", weights_only=False)          # our own file
    assert ck["epoch"] == 1 and math.isfinite(ck["train_stats"]["loss"]) and 0.0 <= ck["val_stats"]["pck"] <= 1.0
    assert len(ck["model"]) == 751
    main(parse(["--epochs", "3", "--resume", cks[-1]]))
    assert os.path.exists(tmp_path / "checkpoint_e002_lr1e-04_bs2_acc2_qpe2.pth")
