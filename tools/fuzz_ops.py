#!/usr/bin/env python
"""Long-running randomised parity run of the HIP operators against the C oracle (bit-exact graph / aggregation /
argmax, backward within rounding) — the same generator as tests/test_hip_ops.py::test_fuzz_random_shapes_bit_exact,
more problems, wider ranges, several seeds.   python tools/fuzz_ops.py --seconds 240 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    rng = np.random.RandomState(args.seed)
    dev = lambda a, dt=torch.float32: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    t0, n = time.time(), 0
    while time.time() - t0 < args.seconds:
        BG = int(rng.randint(1, 7))
        c = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 24, 33, 40, 48, 64, 80, 100, 160, 200]))
        N = int(rng.choice([rng.randint(1, 70), rng.randint(1, 400), rng.randint(300, 1400)]))
        self_graph = rng.rand() < 0.4
        M = N if self_graph else int(rng.choice([rng.randint(1, 70), rng.randint(1, 500), rng.randint(400, 3000)]))
        kd_max = min(M, 64)
        d = int(rng.randint(1, 5))
        k = int(rng.randint(1, max(2, kd_max // d + 1)))
        if k * d > kd_max:
            d, k = 1, min(k, kd_max)
        use_rp = rng.rand() < 0.5
        bf16 = rng.rand() < 0.25
        normalize = rng.rand() < 0.85
        kind = rng.choice(["normal", "normal", "dup", "int"])
        if kind == "int":
            x = rng.randint(-2, 3, size=(BG, c, N)).astype(np.float32)
            y = None if self_graph else rng.randint(-2, 3, size=(BG, c, M)).astype(np.float32)
        else:
            x = (rng.standard_normal((BG, c, N)) * rng.choice([1e-3, 1.0, 30.0])).astype(np.float32)
            y = None if self_graph else (rng.standard_normal((BG, c, M)) * rng.choice([1e-3, 1.0, 30.0])).astype(np.float32)
            if kind == "dup":
                t = x if self_graph else y
                T = t.shape[2] // 3
                if T:
                    t[:, :, T:2 * T] = t[:, :, :T]
                    t[:, :, 2 * T:3 * T] = t[:, :, :T]
        rp = None
        if use_rp:
            rp = -rng.random_sample((N, M)).astype(np.float32)
            if kind != "normal":
                rp = (np.round(rp * 4) / 4).astype(np.float32)
        tdt = torch.bfloat16 if bf16 else torch.float32
        xd = torch.from_numpy(x).to(tdt)
        yd = None if y is None else torch.from_numpy(y).to(tdt)
        xo = xd.float().numpy()
        yo = None if yd is None else yd.float().numpy()
        tag = (n, BG, c, N, M, k, d, use_rp, bf16, normalize, kind)
        want_idx, want_center = O.knn(xo, yo, rp, k, d, normalize=normalize)
        edge = ops.knn_graph(xd.cuda(), None if yd is None else yd.cuda(), None if rp is None else dev(rp).unsqueeze(0), k, d,
                             normalize)
        got = edge.cpu().numpy()
        assert np.array_equal(got[0], want_idx), ("knn idx", tag)
        assert np.array_equal(got[1], want_center), ("knn center", tag)
        want_m, want_arg = O.mr_fwd(xo, yo, want_idx)
        xg = xd.cuda().requires_grad_(True)
        yg = None if yd is None else yd.cuda().requires_grad_(True)
        m = ops.max_relative(xg, edge[0], yg)
        ref_m = torch.from_numpy(want_m).to(tdt).float().numpy()
        assert np.array_equal(m.detach().float().cpu().numpy(), ref_m), ("mr fwd", tag)
        if not bf16:
            g = rng.standard_normal(want_m.shape).astype(np.float32)
            m.backward(dev(g))
            gx, gs = O.mr_bwd(g, want_idx, want_arg, None if y is None else M)
            assert np.allclose(xg.grad.cpu().numpy(), gx, atol=1e-4, rtol=1e-4), ("mr bwd x", tag)
            if y is not None:
                assert np.allclose(yg.grad.cpu().numpy(), gs, atol=1e-4, rtol=1e-4), ("mr bwd src", tag)
        n += 1
    print(f"fuzz seed {args.seed}: {n} random problems bit-exact in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
