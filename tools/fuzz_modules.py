#!/usr/bin/env python
"""Randomised module-level cross-check on the GPU: the fused token-major Grapher / GrapherLabel path against the
composable per-op path (torch conv + BN + the channel-major HIP operators) on random shapes, train mode, forward and
backward.  The composable run replays the graphs the fused run built (the two paths compute their projections with
different GEMM code, so fp32 near-tie neighbours could differ otherwise); what is left is rounding and the rare
near-tie inside max_k(x_j - x_i): agreement = relative L2 error < 2e-3 and > 99.9 % of the elements within 1e-3.   python tools/fuzz_modules.py --seconds 180 --seed 1"""
import argparse, os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")        # the composable path's convs: no exhaustive search per random shape
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


_GRAPHS = []


def run(mod, inputs, cots, fused_on):
    """fused_on: run the fused path and RECORD the graphs it builds; otherwise run the composable path and REPLAY
    them, so that both paths aggregate over identical neighbours and can be compared strictly."""
    from gkgnet_amd import fused, graph
    fused.ENABLED = fused_on
    real_tm, real_fwd = fused.knn_graph_tm, graph.DenseDilatedKnnGraph.forward
    if fused_on:
        del _GRAPHS[:]
        fused.knn_graph_tm = lambda *a, **k: (_GRAPHS.append(real_tm(*a, **k)), _GRAPHS[-1])[1]
    else:
        graph.DenseDilatedKnnGraph.forward = lambda self, x, y=None, relative_pos=None: _GRAPHS.pop(0)
    try:
        return _run(mod, inputs, cots)
    finally:
        fused.knn_graph_tm, graph.DenseDilatedKnnGraph.forward = real_tm, real_fwd


def _run(mod, inputs, cots):
    for p in mod.parameters():
        p.grad = None
    ins = [t.clone().requires_grad_(True) for t in inputs]
    out = mod(*ins)
    outs = [out] if torch.is_tensor(out) else [out[0]]
    torch.autograd.backward(outs, cots[:len(outs)])
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in mod.named_parameters()}
    bufs = {n: b.clone() for n, b in mod.named_buffers() if "running" in n}
    return outs[0].detach(), [t.grad for t in ins], grads, bufs


EVENTS = [0]


def close(a, b, what, tag, strict=False):
    """strict (forward values): > 99.9 % of the elements within 1e-3 and relative L2 error < 2e-3.  Gradients: a
    near-tie inside max_k(x_j - x_i) (the two paths' projections differ by ~1e-6) legitimately re-routes single
    gradient elements, which the dense backward then spreads over a token's channels -> counted as an event and
    held to 95 % / 2e-2."""
    err = (a - b).norm().item() / max(b.norm().item(), 1e-6)
    frac = ((a - b).abs() <= 1e-3 + 1e-3 * b.abs()).float().mean().item()
    small = (a - b).abs().max().item() < 1e-3          # e.g. d(fc1 BN bias): mathematically zero, pure rounding noise
    clean = frac > 0.999 and (err < 2e-3 or small)
    if not clean and not strict:
        EVENTS[0] += 1
        assert frac > 0.95 and err < 2e-2, (what, err, frac, tag)
    else:
        assert clean, (what, err, frac, tag)
    return 0.0 if small else err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    os.environ.setdefault("GKG_RELPOS_DEVICE", "cuda")
    from gkgnet_amd import fused, layers
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    layers.norm_cfg["type"] = "BN"
    rng = np.random.RandomState(args.seed)
    torch.manual_seed(args.seed)
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < args.seconds:
        G = int(rng.choice([1, 2, 4, 8]))
        C = 16 * G * int(rng.randint(1, 5)) if rng.rand() < 0.7 else 16 * int(rng.randint(1, 12))
        if C % G or (C // G) % 4:
            continue
        r = int(rng.choice([1, 1, 2, 4]))
        H = r * int(rng.randint(2, 9)) if r > 1 else int(rng.randint(3, 20))
        W = r * int(rng.randint(2, 9)) if r > 1 else int(rng.randint(3, 20))
        N, M = H * W, H * W // (r * r)
        d = int(rng.randint(1, 4))
        k = int(rng.randint(1, 10))
        if k * d > M:
            k, d = min(k, M), 1
        B = int(rng.randint(1, 6))
        if B * H * W < 32:
            continue
        relpos = bool(rng.rand() < 0.6) and H == W
        tag = dict(B=B, C=C, H=H, W=W, G=G, k=k, d=d, r=r, relpos=relpos)
        g = Grapher(C, k, d, "mr", "gelu", "batch", True, False, 0.2, r, n=N, relative_pos=relpos,
                    use_multi_group=G > 1, num_group=G).cuda().train()
        with torch.no_grad():
            for p in g.parameters():
                if p.dim() == 1:
                    p.uniform_(0.5, 1.5) if p.requires_grad else None
        x = torch.randn(B, C, H, W, device="cuda")
        cot = torch.randn(B, C, H, W, device="cuda")
        fused.ENABLED = True
        if not fused.fused_supported(g, x, G if G > 1 else 1):
            continue
        state = {k_: v.clone() for k_, v in g.state_dict().items()}
        a = run(g, [x], [cot], True)
        g.load_state_dict(state)
        b = run(g, [x], [cot], False)
        worst = max(worst, close(a[0], b[0], "out", tag, strict=True), close(a[1][0], b[1][0], "dx", tag))
        for name in a[2]:
            if name.endswith(".0.bias"):
                continue                      # conv bias in front of train-mode BN: exactly zero on the fused path
            close(a[2][name], b[2][name], "d" + name, tag)
        for name in a[3]:
            assert torch.allclose(a[3][name], b[3][name], atol=1e-4, rtol=1e-3), (name, tag)
        # label module on the same features
        L = int(rng.randint(max(1, -(-16 // B)), 40))       # >= 16 rows per BN batch: fewer make train-mode BN ill-conditioned
        kl = min(int(rng.randint(1, 10)), N)
        gl = GrapherLabel(C, kl, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=N, num_nodes=L,
                          use_multi_group=G > 1, num_group=G).cuda().train()
        e = torch.randn(B, L, C, device="cuda")
        ce = torch.randn(B, L, C, device="cuda")
        state = {k_: v.clone() for k_, v in gl.state_dict().items()}
        a = run(gl, [e, x], [ce], True)
        gl.load_state_dict(state)
        b = run(gl, [e, x], [ce], False)
        tag["L"], tag["kl"] = L, kl
        close(a[0], b[0], "labels", tag, strict=True)
        close(a[1][0], b[1][0], "de", tag)
        close(a[1][1], b[1][1], "dfeat", tag)
        n += 1
        if n % 10 == 0:
            print(f"  {n} configs, {time.time() - t0:.0f}s", flush=True)
    fused.ENABLED = True
    print(f"fuzz_modules seed {args.seed}: {n} random Grapher+GrapherLabel configs agree (worst rel. L2 error {worst:.2e}, "
          f"{EVENTS[0]} gradient tensors touched by a max near-tie) "
          f"in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
