#!/usr/bin/env python
"""BASELINE configs 3 and 5 smoke/timing: full-backbone forward under bf16 autocast with every graph layer on the
HIP kernels.   python tools/run_configs.py cfg3|cfg5 [--batch B]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", choices=["cfg3", "cfg5"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-tune", action="store_true", help="keep the GEMM library's default kernel selection")
    args = ap.parse_args()
    if not args.no_tune:                 # as in bench.py: let TunableOp pick the library GEMM per shape during the first step
        import torch.cuda.tunable as tunable
        tunable.enable(True); tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(20)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gkg_tunableop_{args.cfg}.csv"))
    os.environ.setdefault("GKG_RELPOS_DEVICE", "cuda")
    from gkgnet_amd import _lib, layers
    from gkgnet_amd.backbone import GKGNet
    layers.norm_cfg["type"] = "BN"
    if args.cfg == "cfg3":
        kw, size, B = dict(choice="s", k=9, k_label_gcn=9, n_classes=80, size=576), 576, args.batch or 32
    else:
        kw, size, B = dict(choice="m", k=18, k_label_gcn=18, n_classes=80, size=768, num_group=8), 768, args.batch or 16
    t0 = time.time()
    net = GKGNet(**kw).cuda().eval()
    print(f"built in {time.time() - t0:.1f}s", flush=True)
    img = torch.randn(B, 3, size, size, device="cuda")
    for it in range(args.steps):
        if it == args.steps - 1:
            _lib.prof_reset(); _lib.prof_enable(True)
        torch.cuda.synchronize(); t0 = time.time()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            labels, gap, edge = net(img)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"step {it}: {1e3 * dt:.1f} ms ({B / dt:.1f} img/s) labels {tuple(labels.shape)} {labels.dtype} finite "
              f"{bool(torch.isfinite(labels.float()).all())} edge {tuple(edge.shape)} mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    _lib.prof_enable(False)
    print({k: (round(ms, 2), n) for k, (ms, n) in _lib.prof_read().items() if n})

if __name__ == "__main__":
    main()
