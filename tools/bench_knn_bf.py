#!/usr/bin/env python
"""k-NN kernel time under the bf16-contraction flag (the cfg3 / cfg5 inference form) per stage shape, with and without
relative_pos — library-side HIP-event timing.    python tools/bench_knn_bf.py [shape ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib

SHAPES = {   # name: (BG, c, N, M(None=self), k, d, relpos)
    "s1": (64, 40, 20736, 1296, 9, 1, True), "s1_norp": (64, 40, 20736, 1296, 9, 1, False),
    "s2": (64, 80, 5184, 1296, 9, 1, True), "s2_norp": (64, 80, 5184, 1296, 9, 1, False),
    "s3_d2": (64, 200, 1296, None, 9, 2, True), "s3_d3": (64, 200, 1296, None, 9, 3, True),
    "m1": (128, 12, 36864, 2304, 18, 1, True), "m1_norp": (128, 12, 36864, 2304, 18, 1, False),
    "m2": (128, 24, 9216, 2304, 18, 1, True), "m3_d2": (128, 48, 2304, None, 18, 2, True),
    "m3_d2_norp": (128, 48, 2304, None, 18, 2, False), "m4_d2": (128, 96, 576, None, 18, 2, True),
}


def run(name, iters=5):
    BG, c, N, M, k, d, has_rp = SHAPES[name]
    lib = _lib.load()
    torch.manual_seed(0)
    x = torch.randn(BG, c, N, device="cuda")
    y = None if M is None else torch.randn(BG, c, M, device="cuda")
    Mk = N if M is None else M
    rp = -torch.rand(N, Mk, device="cuda") if has_rp else None
    flags = _lib.KNN_NORMALIZE | _lib.KNN_BF16_CONTRACT | _lib.knn_select_flags()
    edge = torch.empty((1, BG, N, k), dtype=torch.int64, device="cuda")
    ws = torch.empty(int(lib.gkg_knn_workspace_bytes(BG, c, N, Mk, k, d, _lib.F32, flags)), dtype=torch.uint8, device="cuda")

    def call():
        _lib.check(lib.gkg_knn_fwd(x.data_ptr(), None if y is None else y.data_ptr(), None if rp is None else rp.data_ptr(),
                                   edge.data_ptr(), None, BG, c, N, Mk, k, d, _lib.F32, flags, ws.data_ptr(), ws.numel(), None),
                   "gkg_knn_fwd")
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    out = {"shape": name}
    for kn, (ms, cnt) in _lib.prof_read().items():
        if cnt:
            out[kn + "_us"] = round(1e3 * ms / iters, 1)
    rp_gb = (4.0 * N * Mk * BG / 1e9) if has_rp else 0.0
    out["relpos_reads_GB"] = round(rp_gb, 2)
    if has_rp and "knn_tile_us" in out:
        out["relpos_TBs"] = round(rp_gb / out["knn_tile_us"] * 1e3, 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(SHAPES)):
        run(n)
