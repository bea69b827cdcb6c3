"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry
points include/aim_kernels.h declares (no compute is launched without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "aim_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aim_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_table():
    from aim_amd.lib import SIGNATURES
    assert _declared() == sorted(SIGNATURES)


def test_library_exports_every_declared_symbol():
    import aim_amd
    path = aim_amd.library_path()
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), name
    from aim_amd.lib import ABI_VERSION
    assert aim_amd.load_library().aim_version() == ABI_VERSION


def test_missing_library_is_loud(monkeypatch):
    from aim_amd import lib as L
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setattr(L, "library_path", lambda: "/nonexistent/libaim_hip.so")
    with pytest.raises(L.LibraryNotBuilt):
        L.load_library()


def test_ops_refuse_cpu_tensors():
    import torch
    from aim_amd import ops
    a = torch.zeros((8, 8), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(a, a, ops.EPI_BF16, torch.zeros((8, 8), dtype=torch.bfloat16))
