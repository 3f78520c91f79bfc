"""ctypes binding of libgkg_hip.so (C ABI: include/gkg_hip.h).

The product has NO CPU fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GKG_HIP_LIB") or os.path.join(PKG, "libgkg_hip.so")   # GKG_HIP_LIB: same-box A/B of two builds (tools)

ABI_VERSION = 9
F32, BF16, F16 = 0, 1, 2
KNN_NORMALIZE = 1
KNN_BF16_CONTRACT = 2
KNN_SELECT_DIRECT = 4
KNN_SELECT_BUFFERED = 8
KNN_NO_PREFILTER = 16
KNN_FORCE_PREFILTER = 32
KNN_RELPOS_UNIT = 64
KNN_X_PREPARED = 128
KNN_Y_PREPARED = 256


_RELPOS_WARNED = False


def relpos_flags(rp) -> int:
    """KNN_RELPOS_UNIT when every |relative_pos| <= 1.125 (the prefilter kernel's precondition, include/gkg_hip.h).  The check
    is one reduction + a host read, cached ON the tensor object (keyed on its version counter): a module's frozen
    ``relative_pos`` parameter pays it once, in the eager warm-up; a tensor first seen inside a hipGraph capture is not
    vouched for (the call then takes the fp32 tile kernel: same results)."""
    if rp is None:
        return 0
    import torch
    ent = getattr(rp, "_gkg_unit", None)
    if ent is None or ent[0] != rp._version:
        if torch.cuda.is_current_stream_capturing():
            # the range check needs a host read, which a capture cannot make: this call takes the fp32 tile kernel (same graphs,
            # slower for long key streams).  Said once — a user who captures on step 0 loses the prefilter otherwise unnoticed.
            global _RELPOS_WARNED
            if not _RELPOS_WARNED:
                _RELPOS_WARNED = True
                import warnings
                warnings.warn("gkgnet_amd: a relative_pos tensor was first seen inside a hipGraph capture; its value range cannot be "
                              "checked there, so the k-NN of this capture runs without the bf16 prefilter (identical graphs, slower "
                              "at long key streams).  Run one eager warm-up forward before capturing.", RuntimeWarning, stacklevel=3)
            return 0
        ent = (rp._version, bool((rp.detach().abs().max() <= 1.125).item()))
        try:
            rp._gkg_unit = ent
        except AttributeError:
            pass
    return KNN_RELPOS_UNIT if ent[1] else 0


def knn_select_flags() -> int:
    """GKG_KNN_SELECT=direct|buffered, GKG_KNN_PREFILTER=0|force (measurement / tests) -> the C API's mode flags; read here,
    per call, so the library's launch path never calls getenv."""
    sel = os.environ.get("GKG_KNN_SELECT", "")
    f = KNN_SELECT_BUFFERED if sel[:1] == "b" else (KNN_SELECT_DIRECT if sel[:1] == "d" else 0)
    pf = os.environ.get("GKG_KNN_PREFILTER", "")
    if pf == "0" or f:                                             # a forced selection mode means the fp32 tile kernel
        f |= KNN_NO_PREFILTER
    elif pf == "force":
        f |= KNN_FORCE_PREFILTER
    return f
MR_DETERMINISTIC = 1
MR_FP32_ATOMICS = 2
X6_NO_KS, X6_FORCE_KS = 1, 2

EXPORTS = ("gkg_version", "gkg_last_error_string", "gkg_knn_workspace_bytes", "gkg_knn_fwd", "gkg_mr_fwd",
           "gkg_mr_bwd", "gkg_prof_enable", "gkg_prof_reset", "gkg_prof_read", "gkg_prof_work", "gkg_knn_fwd_tm", "gkg_mr_fwd_tm",
           "gkg_mr_bwd_tm", "gkg_nchw_to_tm", "gkg_tm_affine_to_nchw", "gkg_bn_workspace_bytes", "gkg_bn_train_stats",
           "gkg_bn_eval_affine", "gkg_affine_act", "gkg_bn_bwd", "gkg_bn_stats_sums", "gkg_bn_finalize",
           "gkg_bn_bwd_sums", "gkg_bn_bwd_apply", "gkg_linear_stats_doubles",
           "gkg_affine_act_dual", "gkg_edge_stats", "gkg_edge_fwd", "gkg_edge_bwd_stats", "gkg_edge_bwd", "gkg_stream_capture_id", "gkg_x6_planes_bytes", "gkg_x6_prep_desc_bytes",
           "gkg_x6_prep_desc_fill", "gkg_x6_prep_weights", "gkg_linear_bn_fwd_x6", "gkg_linear_dgrad_x6", "gkg_linear_wgrad_x6",
           "gkg_mr_linear_planes_bytes", "gkg_mr_linear_bf16", "gkg_bn_bwd_atomic", "gkg_bn_apply_train",
           "gkg_avgpool_tm", "gkg_bn_apply_knn_prep", "gkg_linear_dgrad_x6_bnbwd",
           "gkg_bn_bwd_apply_from_sums", "gkg_stem_conv3x3s2_supported", "gkg_stem_conv3x3s2_fwd", "gkg_affine_act_bf16in", "gkg_bn_bwd_atomic_scaled",
           "gkg_linear_wgrad_x6_batch", "gkg_x6_splitk_workspace_bytes", "gkg_linear_bn_fwd_x6_sk", "gkg_linear_dgrad_x6_sk",
           "gkg_tm_affine_to_nchw_dual", "gkg_nchw_to_tm_add", "gkg_bn_apply_train_dual",
           "gkg_knn_mr_fused_supported", "gkg_knn_mr_fwd_tm", "gkg_x6_prep_weights_zero",
           "gkg_knn_fwd_tm16", "gkg_mr_fwd_tm16", "gkg_mr_linear_bf16_nn16",
           "gkg_grapher_fwd", "gkg_grapher_bwd", "gkg_grapher_label_fwd", "gkg_grapher_label_bwd")
PROF_KERNELS = ("token_prep", "knn_tile", "knn_merge", "mr_fwd", "mr_bwd", "gemm_x6")

_lib = None


class WgradProblem(C.Structure):
    """include/gkg_hip.h GkgWgradProblem"""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("g_bstride", C.c_size_t), ("x_bstride", C.c_size_t),
                ("ldg", C.c_int), ("ldx", C.c_int), ("R", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("nb", C.c_int),
                ("kperm", C.c_int)]


class GkgError(RuntimeError):
    pass


def load():
    """Loads the HIP library (once).  Raises GkgError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must own the process's HIP runtime: import it BEFORE dlopen-ing our library so that our
    # DT_NEEDED libamdhip64 resolves to the runtime torch already loaded (two runtimes in one process
    # -> "no ROCm-capable device" on the second one).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GkgError(f"{LIB_PATH} not found: build it with `python -m gkgnet_amd._build` "
                       "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.gkg_version.restype = C.c_int
    lib.gkg_last_error_string.restype = C.c_char_p
    lib.gkg_knn_workspace_bytes.restype = C.c_size_t
    lib.gkg_knn_workspace_bytes.argtypes = [C.c_int] * 7 + [C.c_uint]
    lib.gkg_knn_fwd.restype = C.c_int
    lib.gkg_knn_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gkg_mr_fwd.restype = C.c_int
    lib.gkg_mr_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]
    lib.gkg_mr_bwd.restype = C.c_int
    lib.gkg_mr_bwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]
    V, I, Z, F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    lib.gkg_knn_fwd_tm.restype = I
    lib.gkg_knn_fwd_tm.argtypes = [V, I, I] + [V] * 4 + [I] * 8 + [C.c_uint, V, Z, V]
    lib.gkg_knn_mr_fused_supported.restype = I
    lib.gkg_knn_mr_fused_supported.argtypes = [I] * 9 + [C.c_uint]
    lib.gkg_knn_mr_fwd_tm.restype = I
    lib.gkg_knn_mr_fwd_tm.argtypes = [V, I, I] + [V] * 7 + [I] * 7 + [C.c_uint, V, Z, V]
    lib.gkg_mr_fwd_tm.restype = I
    lib.gkg_mr_fwd_tm.argtypes = [V, I, I] + [V] * 4 + [I] * 9 + [V]
    lib.gkg_mr_fwd_tm16.restype = I
    lib.gkg_mr_fwd_tm16.argtypes = [V, I, I] + [V] * 4 + [I] * 9 + [V]
    lib.gkg_bn_apply_knn_prep.restype = I
    lib.gkg_bn_apply_knn_prep.argtypes = [V] * 13 + [I] * 11 + [C.c_uint, I, I, V, V, V, Z, F, F, V, Z, V]
    lib.gkg_avgpool_tm.restype = I
    lib.gkg_avgpool_tm.argtypes = [V, I, I, V, I, I, I, I, I, V]
    lib.gkg_knn_fwd_tm16.restype = I
    lib.gkg_knn_fwd_tm16.argtypes = [V, I, I] + [V] * 3 + [I] * 8 + [C.c_uint, V, Z, V]
    lib.gkg_mr_bwd_tm.restype = I
    lib.gkg_mr_bwd_tm.argtypes = [V] * 5 + [I] * 8 + [C.c_uint, V]
    lib.gkg_nchw_to_tm.restype = I
    lib.gkg_nchw_to_tm.argtypes = [V, V, I, I, I, I, V, V]
    lib.gkg_tm_affine_to_nchw.restype = I
    lib.gkg_tm_affine_to_nchw.argtypes = [V] * 5 + [I, I, I, V, V]
    lib.gkg_tm_affine_to_nchw_dual.restype = I
    lib.gkg_tm_affine_to_nchw_dual.argtypes = [V] * 6 + [I, I, I, V]
    lib.gkg_nchw_to_tm_add.restype = I
    lib.gkg_nchw_to_tm_add.argtypes = [V, V, V, I, I, I, V]
    lib.gkg_bn_apply_train_dual.restype = I
    lib.gkg_bn_apply_train_dual.argtypes = [V] * 15 + [I, I, I, F, F, V, Z, V]
    lib.gkg_bn_workspace_bytes.restype = Z
    lib.gkg_bn_workspace_bytes.argtypes = [I, I, I]
    lib.gkg_bn_train_stats.restype = I
    lib.gkg_bn_train_stats.argtypes = [V] * 10 + [I, I, I, F, F, V, V, Z, V]
    lib.gkg_bn_eval_affine.restype = I
    lib.gkg_bn_eval_affine.argtypes = [V] * 7 + [I, F, V]
    lib.gkg_affine_act.restype = I
    lib.gkg_affine_act.argtypes = [V] * 5 + [I, I, I, I, Z, I, I, I, V, I, V]
    lib.gkg_bn_bwd.restype = I
    lib.gkg_bn_bwd.argtypes = [V] * 9 + [I, I, I, I, Z, I, V, Z, V]
    lib.gkg_bn_apply_train.restype = I
    lib.gkg_bn_apply_train.argtypes = [V] * 14 + [I, I, I, I, Z, I, I, I, V, I, F, F, V, Z, V]
    lib.gkg_bn_bwd_atomic.restype = I
    lib.gkg_bn_bwd_atomic.argtypes = [V] * 9 + [I, I, I, I, Z, I, V, V, Z, V]
    lib.gkg_bn_bwd_atomic_scaled.restype = I
    lib.gkg_bn_bwd_atomic_scaled.argtypes = [V] * 9 + [I, I, I, I, Z, I, V, V, Z, V, I, V]
    lib.gkg_bn_bwd_apply_from_sums.restype = I
    lib.gkg_bn_bwd_apply_from_sums.argtypes = [V] * 9 + [I, I, I, I, Z, I, V, V, Z, V]
    lib.gkg_linear_dgrad_x6_bnbwd.restype = I
    lib.gkg_linear_dgrad_x6_bnbwd.argtypes = [V, I, V, V, I, I, I] + [V] * 6 + [I, I, I, V]
    lib.gkg_bn_stats_sums.restype = I
    lib.gkg_bn_stats_sums.argtypes = [V, V, I, I, I, V, Z, V]
    lib.gkg_bn_finalize.restype = I
    lib.gkg_bn_finalize.argtypes = [V] * 11 + [I, I, F, F, V, V]
    lib.gkg_bn_bwd_sums.restype = I
    lib.gkg_bn_bwd_sums.argtypes = [V] * 10 + [I, I, I, I, Z, I, V, Z, V]
    lib.gkg_bn_bwd_apply.restype = I
    lib.gkg_bn_bwd_apply.argtypes = [V] * 9 + [I, I, I, I, Z, I, V]
    lib.gkg_linear_stats_doubles.restype = I
    lib.gkg_linear_stats_doubles.argtypes = []
    lib.gkg_affine_act_bf16in.restype = I
    lib.gkg_affine_act_bf16in.argtypes = [V, V, V, V, V, I, I, I, V]
    lib.gkg_affine_act_dual.restype = I
    lib.gkg_affine_act_dual.argtypes = [V, V, V, V, V, V, I, I, I, V, I, V]
    lib.gkg_edge_stats.restype = I
    lib.gkg_edge_stats.argtypes = [V, V, V, V, I, I, I, I, I, V]
    lib.gkg_edge_fwd.restype = I
    lib.gkg_edge_fwd.argtypes = [V, V, V, V, V, V, V, I, I, I, I, I, I, V]
    lib.gkg_edge_bwd_stats.restype = I
    lib.gkg_edge_bwd_stats.argtypes = [V] * 10 + [I, I, I, I, I, I, V]
    lib.gkg_edge_bwd.restype = I
    lib.gkg_edge_bwd.argtypes = [V] * 13 + [I, I, I, I, I, I, V]
    lib.gkg_stream_capture_id.restype = C.c_ulonglong
    lib.gkg_stream_capture_id.argtypes = [V]
    lib.gkg_x6_planes_bytes.restype = Z
    lib.gkg_x6_planes_bytes.argtypes = [I, I, I, I]
    lib.gkg_x6_prep_desc_bytes.restype = I
    lib.gkg_x6_prep_desc_bytes.argtypes = []
    lib.gkg_x6_prep_desc_fill.restype = C.c_longlong
    lib.gkg_x6_prep_desc_fill.argtypes = [V, I, V, V, V, I, I, I, C.c_longlong, I]
    lib.gkg_x6_prep_weights.restype = I
    lib.gkg_x6_prep_weights.argtypes = [V, I, C.c_longlong, V]
    lib.gkg_x6_prep_weights_zero.restype = I
    lib.gkg_x6_prep_weights_zero.argtypes = [V, I, C.c_longlong, V, Z, V, Z, V]
    lib.gkg_linear_bn_fwd_x6.restype = I
    lib.gkg_linear_bn_fwd_x6.argtypes = [V, I, Z, V, V, I, I, I, I, I] + [V] * 10 + [F, F, V, V]
    lib.gkg_linear_dgrad_x6.restype = I
    lib.gkg_linear_dgrad_x6.argtypes = [V, I, Z, V, V, I, I, I, I, V]
    lib.gkg_linear_wgrad_x6.restype = I
    lib.gkg_linear_wgrad_x6.argtypes = [V, I, Z, V, I, Z, V, I, I, I, I, I, V]
    lib.gkg_x6_splitk_workspace_bytes.restype = Z
    lib.gkg_x6_splitk_workspace_bytes.argtypes = []
    lib.gkg_linear_bn_fwd_x6_sk.restype = I
    lib.gkg_linear_bn_fwd_x6_sk.argtypes = [V, I, Z, V, V, I, I, I, I, I] + [V] * 10 + [F, F, V, V, Z, C.c_uint, V]
    lib.gkg_linear_dgrad_x6_sk.restype = I
    lib.gkg_linear_dgrad_x6_sk.argtypes = [V, I, Z, V, V, I, I, I, I, V, V, Z, I, Z, C.c_uint, V]
    lib.gkg_linear_wgrad_x6_batch.restype = I
    lib.gkg_linear_wgrad_x6_batch.argtypes = [C.POINTER(WgradProblem), I, I, V]
    lib.gkg_mr_linear_planes_bytes.restype = Z
    lib.gkg_mr_linear_planes_bytes.argtypes = [I]
    lib.gkg_mr_linear_bf16.restype = I
    lib.gkg_mr_linear_bf16.argtypes = [V] * 7 + [I] * 8 + [V]
    lib.gkg_mr_linear_bf16_nn16.restype = I
    lib.gkg_mr_linear_bf16_nn16.argtypes = [V] * 7 + [I] * 8 + [V]
    lib.gkg_stem_conv3x3s2_supported.restype = I
    lib.gkg_stem_conv3x3s2_supported.argtypes = [I, I]
    lib.gkg_stem_conv3x3s2_fwd.restype = I
    lib.gkg_stem_conv3x3s2_fwd.argtypes = [V] * 6 + [I] * 7 + [V]
    lib.gkg_prof_enable.restype = None
    lib.gkg_prof_enable.argtypes = [C.c_int]
    lib.gkg_prof_reset.restype = None
    lib.gkg_prof_read.restype = C.c_int
    lib.gkg_prof_work.restype = C.c_double
    lib.gkg_prof_work.argtypes = [C.c_int]
    lib.gkg_prof_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    v = lib.gkg_version()
    if v != ABI_VERSION:
        raise GkgError(f"libgkg_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gkg_last_error_string().decode(errors="replace")
        raise GkgError(f"{what} failed (rc={rc}): {msg}")


def prof_enable(on: bool = True):
    load().gkg_prof_enable(1 if on else 0)


def prof_reset():
    load().gkg_prof_reset()


def prof_work(name):
    """Algorithmic flop of the launches counted for ``name`` since the last reset (kernels that report it)."""
    return load().gkg_prof_work(PROF_KERNELS.index(name))


def prof_read():
    """{kernel name: (total_ms, launches)} measured with HIP events on the launch stream."""
    lib = load()
    out = {}
    for i, name in enumerate(PROF_KERNELS):
        ms, cnt = C.c_double(0), C.c_long(0)
        check(lib.gkg_prof_read(i, C.byref(ms), C.byref(cnt)), "gkg_prof_read")
        out[name] = (ms.value, cnt.value)
    return out
