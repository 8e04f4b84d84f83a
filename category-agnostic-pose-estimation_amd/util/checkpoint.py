"""Checkpoint files of the CAPE path: same keys as the reference writes (`train_cape_episodic.py:853-947`:
model / optimizer / lr_scheduler / scaler / epoch / args / train_stats / val_stats / best_pck /
epochs_without_improvement / rng_state / np_rng_state / py_rng_state / cuda_rng_state) plus the device dropout counter
of the HIP kernels (`hip_rng_state`), and a loader that executes nothing from the file.

`load_checkpoint` always uses `torch.load(weights_only=True)`; the few non-tensor types such files hold --
`argparse.Namespace` for `args`, numpy arrays inside `np.random.get_state()` -- are allow-listed by name.  A file that needs
anything else is refused (the error names the offending global)."""
import argparse
import random

import numpy as np
import torch


def _safe_globals():
    g = [argparse.Namespace]
    # what numpy arrays / scalars unpickle through (np.random.get_state() holds a uint32 array); data only, no code
    for mod, name in (("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                      ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
        try:
            m = __import__(mod, fromlist=[name])
            g.append(getattr(m, name))
        except (ImportError, AttributeError):
            pass
    g += [np.ndarray, np.dtype]
    for t in ("UInt32DType", "Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType", "UInt8DType"):
        if hasattr(np.dtypes, t):
            g.append(getattr(np.dtypes, t))
    return g


def load_checkpoint(path, map_location="cpu"):
    """Read a checkpoint written by this package or by the reference without unpickling arbitrary objects."""
    with torch.serialization.safe_globals(_safe_globals()):
        return torch.load(path, map_location=map_location, weights_only=True)


def rng_snapshot(device_rng=None):
    """RNG state of every generator the training loop draws from (host torch / numpy / python, torch's device generators,
    and the (seed, step) counter of the HIP dropout kernels)."""
    snap = {"rng_state": torch.get_rng_state(), "np_rng_state": np.random.get_state(), "py_rng_state": random.getstate()}
    if torch.cuda.is_available():
        snap["cuda_rng_state"] = torch.cuda.get_rng_state_all()
    if device_rng is not None:
        snap["hip_rng_state"] = device_rng.t.detach().cpu().clone()
    return snap


def rng_restore(ck, device_rng=None):
    if "rng_state" in ck:
        torch.set_rng_state(ck["rng_state"].cpu() if isinstance(ck["rng_state"], torch.Tensor) else ck["rng_state"])
    if "np_rng_state" in ck:
        st = ck["np_rng_state"]
        np.random.set_state((st[0], np.asarray(st[1], dtype=np.uint32), int(st[2]), int(st[3]), float(st[4])))
    if "py_rng_state" in ck:
        st = ck["py_rng_state"]
        random.setstate((st[0], tuple(st[1]), st[2]))
    if "cuda_rng_state" in ck and torch.cuda.is_available():
        states = [s.cpu() for s in ck["cuda_rng_state"]]
        if len(states) == torch.cuda.device_count():
            torch.cuda.set_rng_state_all(states)
    if device_rng is not None and "hip_rng_state" in ck:
        device_rng.t.copy_(ck["hip_rng_state"].to(device_rng.t.device))
