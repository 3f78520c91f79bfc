"""TEST INFRASTRUCTURE — ctypes wrapper around oracle/libgkg_oracle.so (see gkg_oracle.c).

numpy in, numpy out; fp32 only (bf16 inputs are widened exactly by the caller).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NORMALIZE = 1


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgkg_oracle.so")
    src = os.path.join(_HERE, "gkg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for fn in ("oracle_knn_fwd", "oracle_mr_fwd", "oracle_mr_bwd"):
            getattr(_LIB, fn).restype = C.c_int
    return _LIB


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(C.POINTER(typ))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def knn(x, y=None, relpos=None, k=9, dilation=1, normalize=True, want_dist=False):
    """x (BG,c,N), y (BG,c,M)|None, relpos (N,M)|(1,N,M)|None -> nn_idx (BG,N,k) int64 [, dist (BG,N,M)]."""
    x = _f32(x); y = _f32(y)
    BG, c, N = x.shape
    M = N if y is None else y.shape[2]
    if relpos is not None:
        relpos = _f32(relpos).reshape(N, M)
    idx = np.empty((BG, N, k), np.int64)
    center = np.empty((BG, N, k), np.int64)
    dist = np.empty((BG, N, M), np.float32) if want_dist else None
    rc = lib().oracle_knn_fwd(_p(x, C.c_float), _p(y, C.c_float), _p(relpos, C.c_float), _p(idx, C.c_int64),
                              _p(center, C.c_int64), BG, c, N, M, k, dilation,
                              C.c_uint(NORMALIZE if normalize else 0), _p(dist, C.c_float))
    if rc != 0:
        raise ValueError(f"oracle_knn_fwd failed: {rc}")
    return (idx, center, dist) if want_dist else (idx, center)


def mr_fwd(x, src, nn_idx):
    x = _f32(x); src = _f32(src)
    BG, c, N = x.shape
    M = N if src is None else src.shape[2]
    nn_idx = np.ascontiguousarray(nn_idx, dtype=np.int64)
    k = nn_idx.shape[2]
    m = np.empty((BG, c, N), np.float32)
    arg = np.empty((BG, c, N), np.uint8)
    rc = lib().oracle_mr_fwd(_p(x, C.c_float), _p(src, C.c_float), _p(nn_idx, C.c_int64), _p(m, C.c_float),
                             _p(arg, C.c_uint8), BG, c, N, M, k)
    if rc != 0:
        raise ValueError(f"oracle_mr_fwd failed: {rc}")
    return m, arg


def mr_bwd(g, nn_idx, argmax, M=None):
    """Returns (gx, gsrc); gsrc is None for the self graph (M None)."""
    g = _f32(g)
    BG, c, N = g.shape
    nn_idx = np.ascontiguousarray(nn_idx, dtype=np.int64)
    argmax = np.ascontiguousarray(argmax, dtype=np.uint8)
    k = nn_idx.shape[2]
    gx = np.empty((BG, c, N), np.float32)
    gsrc = None if M is None else np.empty((BG, c, M), np.float32)
    rc = lib().oracle_mr_bwd(_p(g, C.c_float), _p(nn_idx, C.c_int64), _p(argmax, C.c_uint8), _p(gx, C.c_float),
                             _p(gsrc, C.c_float), BG, c, N, N if M is None else M, k)
    if rc != 0:
        raise ValueError(f"oracle_mr_bwd failed: {rc}")
    return gx, gsrc
