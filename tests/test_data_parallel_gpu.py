"""Two data-parallel ranks of the real model on one GPU (gloo carrying CUDA buckets): the all-reduced mean gradient
equals the locally accumulated one -- buckets launched from hooks and direct-to-arena notifications, side stream,
communication stream, finish(), and the parameter broadcast (ranks start from different seeds)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_local_accumulation():
    sys.path.insert(0, ROOT)
    from tools.ddp_rehearsal import worker
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)       # raises if a rank's assertions fail
