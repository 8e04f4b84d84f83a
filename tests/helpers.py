"""Shared test helpers: build the product model with procedural weights."""
import argparse

import torch


def build_product(extra=(), device="cuda", proc_sd=None):
    import cape_amd  # noqa: F401
    from cape_amd.datasets import DiscreteTokenizerV2
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", *extra])
    tok = DiscreteTokenizerV2(int(args.vocab_size ** 0.5), args.seq_len, add_cls=False)
    base, crit = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base)
    if proc_sd is not None:
        missing, unexpected = model.load_state_dict(proc_sd, strict=True)
        assert not missing and not unexpected
    return args, tok, model.to(device), crit.to(device)


def to_dev(batch, device="cuda"):
    out = dict(batch)
    for k in ("images", "support_coords", "support_mask"):
        out[k] = batch[k].to(device)
    out["targets"] = {k: v.to(device) for k, v in batch["targets"].items()}
    return out
