#!/usr/bin/env python
"""Full GKGNet-576 (pvig_s) forward + backward on one MI355X with every graph layer on the HIP kernels.
Stress / timing run for the large shapes (stage 1: 20 736 queries x 1 296 pooled keys; label graph 80 x 20 736).
    GKG_RELPOS_DEVICE=cuda python tools/run_backbone.py [--batch 4] [--size 576] [--steps 3]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=576)
    ap.add_argument("--choice", default="s")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--classes", type=int, default=80)
    args = ap.parse_args()
    from gkgnet_amd import _lib, layers
    from gkgnet_amd.backbone import GKGNet
    layers.norm_cfg["type"] = "BN"
    t0 = time.time()
    net = GKGNet(choice=args.choice, n_classes=args.classes, size=args.size).cuda().train()
    print(f"built in {time.time() - t0:.1f}s; params {sum(p.numel() for p in net.parameters() if p.requires_grad) / 1e6:.2f} M", flush=True)
    x = torch.randn(args.batch, 3, args.size, args.size, device="cuda")
    for it in range(args.steps):
        if it == args.steps - 1:
            _lib.prof_reset(); _lib.prof_enable(True)
        torch.cuda.synchronize(); t0 = time.time()
        labels, gap, edge = net(x)
        loss = labels.square().mean() + gap.square().mean()
        loss.backward()
        torch.cuda.synchronize()
        print(f"step {it}: {1e3 * (time.time() - t0):.1f} ms  loss {loss.item():.4f} finite {bool(torch.isfinite(labels).all())} "
              f"edge {tuple(edge.shape)} mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
        net.zero_grad(set_to_none=True)
    _lib.prof_enable(False)
    for k, (ms, n) in _lib.prof_read().items():
        if n:
            print(f"  {k:12s} {ms:9.2f} ms in {n} launches")

if __name__ == "__main__":
    main()
