"""Build libgkg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(PKG, "libgkg_hip.so")
SOURCES = ["gkg_api.hip", "gkg_knn.hip", "gkg_knn_f32.hip", "gkg_knn_f32_norp.hip", "gkg_knn_f32_mr.hip", "gkg_knn_f32_mr_norp.hip", "gkg_knn_bf.hip", "gkg_knn_bf_norp.hip", "gkg_knn_pf.hip", "gkg_knn_pf_norp.hip", "gkg_mr.hip", "gkg_dense.hip", "gkg_gemm_x6.hip", "gkg_edge.hip", "gkg_mrgemm.hip", "gkg_stem.hip", "gkg_block.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-pass-failed", "-I" + INCLUDE, "-I" + CSRC] + \
        os.environ.get("GKG_BUILD_FLAGS", "").split()          # measurement builds (-D switches of the ablation tools)


def csrc_sha16() -> str:
    """First 16 hex digits of a SHA-256 over the kernel sources (csrc/*.hip, csrc/*.h, include/gkg_hip.h, sorted by name): the
    identity of the kernels a measurement was taken on, computable wherever the sources are (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(INCLUDE, "gkg_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "gkg_hip.h")]
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    # objects are only as good as the flags they were compiled with (GKG_BUILD_FLAGS adds -D ablation switches: no stores, no
    # loads ...): a stamp of the flag list sits beside them, and a different list rebuilds everything
    stamp = os.path.join(objdir, ".flags")
    flags_now = " ".join(FLAGS)
    try:
        flags_then = open(stamp).read()
    except OSError:
        flags_then = None
    if flags_then != flags_now:
        force = force or flags_then is not None or bool(os.environ.get("GKG_BUILD_FLAGS", "").split())
        with open(stamp, "w") as fh:
            fh.write(flags_now)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        deps = [src] + headers
        with open(src) as fh:                            # a wrapper that #includes a sibling source
            for line in fh:
                if line.startswith('#include "') and line.rstrip().endswith('.hip"'):
                    deps.append(os.path.join(CSRC, line.split('"')[1]))
        if force or _stale(obj, deps):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
