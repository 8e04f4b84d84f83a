"""Does a replayed hipGraph run independent branches (captured from two streams) concurrently on this ROCm build?
Two chains of half-chip kernels: serial time vs two eager streams vs one captured graph with a fork / join."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cape_amd  # noqa
from cape_amd.hip import ops

dev = "cuda"
M, N, K = 6400, 256, 256            # 200 blocks: about half the chip
A = [torch.randn(M, K, device=dev) for _ in range(2)]
W = [torch.randn(N, K, device=dev) for _ in range(2)]
C = [torch.empty(M, N, device=dev) for _ in range(2)]
side = torch.cuda.Stream()
n = 40


def chain(i):
    for _ in range(n):
        ops.gemm(A[i], W[i], C[i], M, N, K)


def both_serial():
    chain(0); chain(1)


def both_forked():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(1)
    chain(0)
    torch.cuda.current_stream().wait_stream(side)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print(f"eager serial      {timeit(both_serial):8.3f} ms")
print(f"eager two streams {timeit(both_forked):8.3f} ms")
for name, fn in (("serial", both_serial), ("forked", both_forked)):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
    print(f"graph {name:7s}     {timeit(g.replay):8.3f} ms")
