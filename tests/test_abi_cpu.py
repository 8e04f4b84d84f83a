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
    assert lib.abi_version() == 6
    assert isinstance(lib.last_error(), str)


def test_null_descriptor_is_rejected_without_gpu():
    import ctypes
    rc = lib.raw().cape_gemm_f32(None, None)
    assert rc != 0 and "null" in lib.last_error()
