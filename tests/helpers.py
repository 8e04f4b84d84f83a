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


def proc_sd_512():
    """Procedural weights of the --image_size 512 model: the 256 spec with the patch-2 input_proj shapes
    (tests/golden/state_dict_spec_512_diff.json, emitted by oracle/make_golden_r2.py)."""
    import json
    import os
    from oracle import procweights
    diff = dict((k, tuple(s)) for k, s in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_spec_512_diff.json"))))
    spec = [(k, diff.get(k, s)) for k, s in procweights.load_spec()]
    return procweights.procedural_state_dict(spec)


def cfg5_episode_batch():
    """The config-5 inputs of cfg5_512_decode.npz: one 5-shot episode, P = 68, two 512x512 queries, through the
    product's collate (bit-equal to the reference's, tests/test_host_contract_cpu.py)."""
    import cape_amd  # noqa: F401
    from cape_amd.datasets import episodic_collate_fn
    from oracle import cape_ref, synth
    return episodic_collate_fn([synth.make_episode(41, 512, 68, 2, 5, cape_ref.Cfg(patch_size=2))])


def train_loop_batches():
    """The three micro-batches of train_loop.npz (64x64, 9 keypoints, one episode x two queries each)."""
    import cape_amd  # noqa: F401
    from cape_amd.datasets import episodic_collate_fn
    from oracle import cape_ref, synth
    cfg = cape_ref.Cfg(dropout=0.0)
    return [episodic_collate_fn([synth.make_episode(60 + i, 64, 9, 2, 1, cfg, category_id=1 + i, n_invisible=2 * (i % 2))])
            for i in range(3)]
