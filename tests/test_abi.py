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


def test_no_kernel_runs_out_of_scratch_memory():
    """the max-relative kernels (gather / scatter, every form) carry NO scratch memory — run-time
    `mode` / `arg_kind` branches once sent the float4s of the scatter's inner loop through it — and no kernel of the library
    spills more than a handful of registers (the 64-entry buffered k-NN forms spilled 100-400 and were removed).  Read from
    the AMDGPU metadata notes of the gfx950 code objects embedded in the shipped library (tools/kernel_meta.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    from gkgnet_amd import _build
    ks = kernel_meta.kernels(_build.build())
    assert len(ks) > 100
    mr = {n: k for n, k in ks.items() if "mr_" in n and "mr_linear" not in n}
    assert len(mr) >= 30, sorted(mr)
    bad = {n: k.get("private_segment_fixed_size") for n, k in mr.items() if k.get("private_segment_fixed_size", 0) != 0}
    assert not bad, bad
    heavy = {n: k.get("vgpr_spill_count") for n, k in ks.items() if k.get("vgpr_spill_count", 0) > 8}
    assert not heavy, heavy


def test_round5_host_side_queries_and_argument_checks():
    """Pure host code of the round-5 entry points: which graph shapes take the fused k-NN + aggregation kernel (the launch plan
    without a launch), the split-K workspace size, and the argument validation of the batched weight gradient."""
    from gkgnet_amd import _build, _lib
    lib = _lib.load()
    f = _lib.KNN_NORMALIZE
    assert lib.gkg_knn_mr_fused_supported(32, 4, 80, 324, 324, 9, 1, 0, 1, f) == 1          # cfg2 Grapher graph
    assert lib.gkg_knn_mr_fused_supported(32, 4, 80, 80, 324, 9, 1, 1, 0, f) == 1           # cfg2 label graph
    assert lib.gkg_knn_mr_fused_supported(2, 4, 80, 324, 324, 9, 1, 0, 1, f) == 0           # 48 workgroups: the keys are split
    assert lib.gkg_knn_mr_fused_supported(32, 2, 40, 20736, 1296, 9, 1, 1, 1, f | _lib.KNN_RELPOS_UNIT) == 0   # prefilter shape
    assert lib.gkg_knn_mr_fused_supported(32, 4, 80, 324, 324, 9, 1, 0, 1, f | _lib.KNN_BF16_CONTRACT) == 0    # bf16 contraction
    assert lib.gkg_knn_mr_fused_supported(0, 4, 80, 324, 324, 9, 1, 0, 1, f) == 0
    ws = lib.gkg_x6_splitk_workspace_bytes()
    assert ws >= 4096 + 512 * 32768 and ws % 16 == 0
    assert lib.gkg_linear_wgrad_x6_batch(None, 0, 0, None) != 0
    assert lib.gkg_linear_wgrad_x6_batch(None, 3, 0, None) != 0 and b"no problems" in lib.gkg_last_error_string()
    h = _build.csrc_sha16()
    assert len(h) == 16 and int(h, 16) >= 0 and h == _build.csrc_sha16()
