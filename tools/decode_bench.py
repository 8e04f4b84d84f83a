"""Timing of the cached autoregressive decode (CAPEModel.forward_inference) on synthetic episodes: the decode loop alone
(HIP events around it, image encoding excluded) per generated step, eager and as replayed per-step hipGraphs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cape_amd  # noqa: E402,F401
from cape_amd.runtime.decode_bench import decode_benchmark  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, nargs="+", default=[1, 16])
    ap.add_argument("--image_size", type=int, default=512)
    ap.add_argument("--keypoints", type=int, default=68)
    ap.add_argument("--shots", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    for e in a.episodes:
        r = decode_benchmark(torch.device("cuda"), episodes=e, image_size=a.image_size, keypoints=a.keypoints, shots=a.shots, reps=a.reps)
        print(r, flush=True)


if __name__ == "__main__":
    main()
