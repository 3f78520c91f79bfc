#!/usr/bin/env python
"""k-NN kernel time on the MODEL'S OWN activations: one forward of a BASELINE backbone config (random-init weights, N(0,1)
images, bf16 autocast) with fused.knn_graph_tm intercepted; every call is then replayed on its captured inputs (5 x, HIP-event
time of the library's k-NN scopes) and on N(0,1) tensors of the same shape.  Selection cost depends on the data: real
activations admit far more candidates than random ones.
    python tools/bench_knn_model_inputs.py [cfg3|cfg5] [flags...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gkgnet_amd import _lib, fused
from gkgnet_amd.backbone import GKGNet


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    spec = bench.BACKBONE_WORKLOADS[wl]
    torch.manual_seed(0)
    net = GKGNet(**dict(spec["kw"])).cuda().eval()
    B = spec["B"]
    img = torch.randn(B, 3, spec["kw"]["size"], spec["kw"]["size"], generator=torch.Generator().manual_seed(100)).cuda()
    calls = []
    orig = fused.knn_graph_tm

    def spy(x, y, rp, k, d, G):
        calls.append((x.detach().clone(), None if y is None else y.detach().clone(), rp, k, d, G))
        return orig(x, y, rp, k, d, G)
    fused.knn_graph_tm = spy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net(img)
        calls.clear()
        net(img)
    fused.knn_graph_tm = orig

    def t(x, y, rp, k, d, G):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(2):
                orig(x, y, rp, k, d, G)
            torch.cuda.synchronize()
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(5):
                orig(x, y, rp, k, d, G)
            torch.cuda.synchronize()
            _lib.prof_enable(False)
        pr = _lib.prof_read()
        return {n: round(v[0] / 5 * 1e3, 1) for n, v in pr.items() if v[1] and n in ("token_prep", "knn_tile", "knn_merge")}
    tot_m = tot_r = 0.0
    for i, (x, y, rp, k, d, G) in enumerate(calls):
        m = t(x, y, rp, k, d, G)
        r = t(torch.randn_like(x), None if y is None else torch.randn_like(y), rp, k, d, G)
        tot_m += m.get("knn_tile", 0); tot_r += r.get("knn_tile", 0)
        if rp is not None and "rpvariants" in sys.argv:
            rr = t(x, y, -torch.rand_like(rp.float()), k, d, G)
            rs = t(x, y, rp.float().flip(-1).contiguous(), k, d, G)
            print(f"         model features with relative_pos replaced by -U(0,1): {rr}; by its own mirror image along the keys: {rs}")
        print(f"call {i:2d}: x {tuple(x.shape)} y {None if y is None else tuple(y.shape)} G={G} k={k} d={d} rp={'yes' if rp is not None else 'no'}: "
              f"model inputs {m}   N(0,1) inputs {r}", flush=True)
    print(f"sum of the k-NN kernel scopes per forward: model inputs {tot_m / 1e3:.2f} ms, N(0,1) inputs {tot_r / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
