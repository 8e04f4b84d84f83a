"""cape_amd -- MI355X-native CAPE episodic training / inference hot path.

Python here is host glue only (module tree with the reference's parameter names, autograd
plumbing, torch.distributed); every arithmetic op of the hot path runs in `csrc/` HIP kernels
through the C ABI of `include/cape_hip.h` (`hip/lib.py` binds it with ctypes).  There is no CPU
fallback: importing `cape_amd.hip.lib` without a built `libcape_hip.so` raises.
"""
__version__ = "0.1.0"
