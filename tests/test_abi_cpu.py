"""CPU checks of the C-ABI boundary: the library loads and exports every symbol that
include/cape_hip.h declares (no compute calls without a GPU)."""
import os
import re

import cape_amd  # noqa: F401
from cape_amd.hip import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cape_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cape_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 40
    raw = lib.raw()
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in cape_hip.h but not exported by libcape_hip.so"


def test_binding_covers_header():
    assert set(declared_symbols()) == set(lib.EXPORTS)


def test_abi_version_and_error_string():
    assert lib.abi_version() == 11
    assert isinstance(lib.last_error(), str)


def test_null_descriptor_is_rejected_without_gpu():
    import ctypes
    rc = lib.raw().cape_gemm_f32(None, None)
    assert rc != 0 and "null" in lib.last_error()


def test_decode_step_descriptor_checks_without_gpu():
    """cape_decode_step validates its descriptor on the host before anything touches a device: null descriptor, a step outside the
    cache, a layer count beyond the table, a sampling layout other than 4 levels x 4 points."""
    import ctypes
    raw = lib.raw()
    assert raw.cape_decode_step(None, None) != 0 and "null" in lib.last_error()
    d = lib.DecodeStepDesc()
    d.N, d.n_layers, d.T, d.step, d.P, d.S, d.L, d.n_points, d.ncls, d.ffn_dim = 2, 6, 40, 40, 17, 100, 4, 4, 3, 1024
    assert raw.cape_decode_step(ctypes.byref(d), None) != 0 and "step" in lib.last_error()
    d.step, d.n_layers = 3, lib.DECODE_MAX_LAYERS + 1
    assert raw.cape_decode_step(ctypes.byref(d), None) != 0 and "layers" in lib.last_error()
    d.n_layers, d.n_points = 6, 2
    assert raw.cape_decode_step(ctypes.byref(d), None) != 0 and "points" in lib.last_error()
    assert ctypes.sizeof(lib.DecodeStepDesc) == 2600 and ctypes.sizeof(lib.DecodeLayerDesc) == 37 * 8
