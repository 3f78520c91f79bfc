"""Host time of the cfg2 step launched eagerly (no hipGraph): wall per step, GPU-busy per step and the Python profile.

    python tools/debug/eager_profile.py [--steps 400] [--workload cfg2] [--top 40]

The replayed step is device-bound (~0.81 ms); launched op by op the same step is bound by the host.  This prints where the
host time goes (cProfile, sorted by own time) so that what is left above the replay can be named function by function."""
import argparse
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench                                                    # noqa: E402
from gkgnet_amd import parallel                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    w = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    grapher, label = bench.build_modules(w, dev)
    params = list(grapher.parameters()) + list(label.parameters())
    bucket = parallel.GradBucket(params)
    B, C, H, L = args.batch, w["C"], w["H"], w["L"]
    x = torch.randn(B, C, H, H, device=dev).requires_grad_(True)
    e = torch.randn(B, L, C, device=dev).requires_grad_(True)
    cx, ce = torch.randn(B, C, H, H, device=dev), torch.randn(B, L, C, device=dev)

    def step():
        bucket.release(prezero=True)
        x.grad = None
        e.grad = None
        out = grapher(x)
        e2, _ = label(e, out)
        torch.autograd.backward([out, e2], [cx, ce])
        bucket.pack()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager: {1e3 * t_all / args.steps:.4f} ms/step wall ({1e3 * t_issue / args.steps:.4f} ms/step to issue)")
    # own timers around the four block calls (forward: this thread; backward: the autograd engine's thread, which cProfile
    # does not see) and around the C entry points inside them
    from gkgnet_amd import _lib, block, fused
    lib = _lib.load()
    acc = {}

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return w
    for nm in ("gkg_grapher_fwd", "gkg_grapher_bwd", "gkg_grapher_label_fwd", "gkg_grapher_label_bwd", "gkg_linear_wgrad_x6_batch",
               "gkg_x6_prep_desc_fill"):
        if hasattr(lib, nm):
            setattr(lib, nm, timed("C " + nm, getattr(lib, nm)))
    for cls, nm in ((block._GrapherBlockFn, "grapher"), (block._LabelBlockFn, "label")):
        cls.forward = staticmethod(timed(f"py {nm}.forward (incl. C)", cls.forward))
        cls.backward = staticmethod(timed(f"py {nm}.backward (incl. C)", cls.backward))
    fused.flush_wgrads = timed("py flush_wgrads (incl. C)", fused.flush_wgrads)
    for nm in ("_proj_fwd", "_proj_bwd", "_graph_op", "_issue_wgrads", "try_grapher", "try_label", "_run_grapher"):
        setattr(block, nm, timed("  block." + nm, getattr(block, nm)))
    block._Plan.valid = timed("  block._Plan.valid", block._Plan.valid)
    fused._grad_outs = timed("  fused._grad_outs", fused._grad_outs)
    fused._label_features = timed("  fused._label_features", fused._label_features)
    bucket.pack = timed("py bucket.pack", bucket.pack)
    bucket.release = timed("py bucket.release", bucket.release)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"with timers: {1e3 * t_issue / args.steps:.4f} ms/step to issue")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {1e6 * v / args.steps:8.1f} us/step  {k}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
