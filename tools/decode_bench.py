"""Timing of the cached autoregressive decode (CAPEModel.forward_inference) on synthetic episodes: ms per generated step."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn  # noqa: E402
from cape_amd.datasets.synthetic import SyntheticEpisodes  # noqa: E402
from cape_amd.models import build_model  # noqa: E402
from cape_amd.models.cape_model import build_cape_model  # noqa: E402
from cape_amd.models.train_cape_episodic import get_args_parser  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=16)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--no_graph", action="store_true")
    a = ap.parse_args()
    args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
        ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", str(a.image_size)])
    torch.manual_seed(0)
    tok = DiscreteTokenizerV2(44, args.seq_len)
    base, _ = build_model(args, tokenizer=tok)
    model = build_cape_model(args, base).cuda().eval()
    ds = SyntheticEpisodes(tok, a.episodes, a.image_size, 17, 2, seed=3)
    b = episodic_collate_fn([ds[j] for j in range(a.episodes)])
    im, sc, sm, sk = b["query_images"].cuda(), b["support_coords"].cuda(), b["support_masks"].cuda(), b["support_skeletons"]
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
    with torch.no_grad():
        for r in range(a.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.forward_inference(im, sc, sm, skeleton_edges=sk, graph=not a.no_graph)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            T = out["logits"].shape[1]
            if True:
                print(f"N={im.shape[0]} images {a.image_size}^2: {T} steps in {dt * 1e3:.1f} ms = {dt / T * 1e3:.3f} ms/step, "
                      f"{im.shape[0] * T / dt:.0f} tokens/s", flush=True)


if __name__ == "__main__":
    main()
