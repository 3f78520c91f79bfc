"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gkg_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gkg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gkg_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from gkgnet_amd import _build, _lib
    so = _build.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gkg_hip.h but not exported"
    lib.gkg_version.restype = ctypes.c_int
    assert lib.gkg_version() == _lib.ABI_VERSION


def test_workspace_query_is_pure_host_code():
    from gkgnet_amd import _lib
    lib = _lib.load()
    n = lib.gkg_knn_workspace_bytes(128, 80, 324, 324, 9, 1, _lib.F32, 1)
    assert n >= 128 * 80 * 324 * 4
    assert lib.gkg_knn_workspace_bytes(1, 8, 16, 16, 9, 2, _lib.F32, 1) == 0     # k*d > M -> unsupported


def test_cpu_tensors_are_rejected_not_routed_to_a_fallback():
    import pytest
    import torch
    from gkgnet_amd import _lib, ops
    with pytest.raises(_lib.GkgError):
        ops.knn_graph(torch.zeros(1, 4, 8), None, None, 3, 1)
    with pytest.raises(_lib.GkgError):
        ops.max_relative(torch.zeros(1, 4, 8), torch.zeros(1, 8, 3, dtype=torch.int64))
