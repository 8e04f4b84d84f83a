"""CLI of the CAPE episodic trainer on MI355X -- same flags and defaults as the reference's
`models/train_cape_episodic.py:86-254` (`get_args_parser`), same checkpoint naming/rotation (:853-947).

    python category-agnostic-pose-estimation_amd/models/train_cape_episodic.py --use_geometric_encoder --use_gcn_preenc ...
    torchrun --nproc-per-node 8 ... (one process per GPU, RCCL gradient all-reduce)

Data: `--dataset_name synthetic` trains on seeded MP-100-shaped episodes (no dataset ships with the
GPU box); the MP-100 file loader of the reference is host I/O outside the hot path."""
import argparse
import os
import sys

import numpy as np

if __package__ in (None, ""):                      # executed as a script: make `cape_amd` importable
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, _root)
    import cape_amd  # noqa: F401

_T, _F = True, False
# (flag, kwargs) -- one row per reference flag, grouped as in the reference parser
_FLAGS = [
    # CAPE
    ("--cape_mode", dict(action="store_true", default=_T)),
    ("--support_encoder_layers", dict(default=3, type=int)),
    ("--support_fusion_method", dict(default="cross_attention", choices=["cross_attention", "concat", "add"])),
    ("--num_queries_per_episode", dict(default=2, type=int)),
    ("--episodes_per_epoch", dict(default=1000, type=int)),
    ("--val_episodes_per_epoch", dict(default=200, type=int)),
    ("--fixed_val_episodes", dict(action="store_true")),
    ("--val_seed", dict(default=42, type=int)),
    ("--category_split_file", dict(default="category_splits.json")),
    ("--use_geometric_encoder", dict(action="store_true", default=_F)),
    ("--use_gcn_preenc", dict(action="store_true", default=_F)),
    ("--num_gcn_layers", dict(default=2, type=int)),
    ("--debug_overfit_category", dict(default=None, type=int)),
    ("--debug_overfit_episodes", dict(default=10, type=int)),
    # optimisation
    ("--lr", dict(default=1e-4, type=float)),
    ("--lr_backbone_names", dict(default=["backbone.0"], type=str, nargs="+")),
    ("--lr_backbone", dict(default=1e-5, type=float)),
    ("--lr_linear_proj_names", dict(default=["sampling_offsets"], type=str, nargs="+")),
    ("--lr_linear_proj_mult", dict(default=0.1, type=float)),
    ("--batch_size", dict(default=2, type=int)),
    ("--accumulation_steps", dict(default=4, type=int)),
    ("--weight_decay", dict(default=1e-4, type=float)),
    ("--epochs", dict(default=300, type=int)),
    ("--lr_drop", dict(default="200,250", type=str)),
    ("--scheduler", dict(default="cosine_warmrestarts", choices=["multistep", "cosine_warmrestarts", "onecycle"])),
    ("--warmup_epochs", dict(default=5, type=int)),
    ("--T_0", dict(default=20, type=int)),
    ("--T_mult", dict(default=2, type=int)),
    ("--eta_min", dict(default=1e-6, type=float)),
    ("--early_stopping_patience", dict(default=20, type=int)),
    ("--clip_max_norm", dict(default=0.1, type=float)),
    # input
    ("--input_channels", dict(default=3, type=int)),
    ("--image_size", dict(default=256, type=int)),
    ("--image_norm", dict(action="store_true")),
    ("--debug", dict(action="store_true")),
    # backbone
    ("--backbone", dict(default="resnet50", type=str)),
    ("--dilation", dict(action="store_true")),
    ("--position_embedding", dict(default="sine", type=str)),
    ("--position_embedding_scale", dict(default=2 * np.pi, type=float)),
    ("--num_feature_levels", dict(default=4, type=int)),
    # transformer
    ("--enc_layers", dict(default=6, type=int)),
    ("--dec_layers", dict(default=6, type=int)),
    ("--dim_feedforward", dict(default=1024, type=int)),
    ("--hidden_dim", dict(default=256, type=int)),
    ("--dropout", dict(default=0.1, type=float)),
    ("--nheads", dict(default=8, type=int)),
    # sequence
    ("--poly2seq", dict(action="store_true", default=_T)),
    ("--num_queries", dict(default=200, type=int)),
    ("--seq_len", dict(default=200, type=int)),
    ("--num_polys", dict(default=1, type=int)),
    ("--vocab_size", dict(default=2000, type=int)),
    ("--masked_attn", dict(action="store_true", default=_F)),
    ("--dec_n_points", dict(default=4, type=int)),
    ("--enc_n_points", dict(default=4, type=int)),
    ("--query_pos_type", dict(default="sine", type=str)),
    ("--with_poly_refine", dict(default=_T, action="store_true")),
    ("--use_anchor", dict(action="store_true")),
    ("--semantic_classes", dict(default=70, type=int)),
    # loss
    ("--no_aux_loss", dict(dest="aux_loss", action="store_false")),
    ("--aux_loss", dict(action="store_true", default=_T)),
    ("--cls_loss_coef", dict(default=1, type=float)),
    ("--coords_loss_coef", dict(default=5, type=float)),
    ("--room_cls_loss_coef", dict(default=0.0, type=float)),
    ("--raster_loss_coef", dict(default=0.0, type=float)),
    ("--eos_weight", dict(default=20.0, type=float)),
    ("--label_smoothing", dict(default=0.0, type=float)),
    # data
    ("--dataset_name", dict(default="mp100", type=str)),
    ("--dataset_root", dict(default=".", type=str)),
    ("--mp100_split", dict(default=1, type=int, choices=[1, 2, 3, 4, 5])),
    # decoder architecture
    ("--dec_layer_type", dict(default="v1", type=str)),
    ("--dec_attn_concat_src", dict(action="store_true")),
    ("--dec_qkv_proj", dict(action="store_true", default=_T)),
    ("--pre_decoder_pos_embed", dict(action="store_true")),
    ("--learnable_dec_pe", dict(action="store_true")),
    ("--add_cls_token", dict(action="store_true", default=_F)),
    ("--inject_cls_embed", dict(action="store_true", default=_F)),
    ("--patch_size", dict(default=1, type=int)),
    ("--freeze_anchor", dict(action="store_true")),
    ("--per_token_sem_loss", dict(action="store_true", default=_F)),
    # run
    ("--output_dir", dict(default="output/cape_episodic")),
    ("--device", dict(default=None)),
    ("--seed", dict(default=42, type=int)),
    ("--resume", dict(default="")),
    ("--start_epoch", dict(default=0, type=int)),
    ("--num_workers", dict(default=2, type=int)),
    ("--job_name", dict(default="cape_episodic", type=str)),
    ("--use_amp", dict(action="store_true")),
    ("--cudnn_benchmark", dict(action="store_true")),
    ("--use_wandb", dict(action="store_true")),
    ("--wandb_project", dict(default="MP100-CAPE-Episodic", type=str)),
    ("--print_freq", dict(default=10, type=int)),
]


def get_args_parser():
    parser = argparse.ArgumentParser("CAPE Episodic Training", add_help=False)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def main(args):
    from cape_amd.models import engine_cape
    return engine_cape.run_training(args)


if __name__ == "__main__":
    parser = argparse.ArgumentParser("CAPE episodic training (MI355X)", parents=[get_args_parser()])
    main(parser.parse_args())
