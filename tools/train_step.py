#!/usr/bin/env python
"""GKGNet-576 training step on synthetic multi-label targets (BASELINE config 4 shape): backbone (all 16 graph layers
on the HIP kernels) + LabelQueryHead + ASL x10 + smoothed BCE, AdamW (no decay on norm/bias), grad-clip 5.0,
data-parallel over the visible ranks with ONE flat RCCL gradient all-reduce per step.

    GKG_RELPOS_DEVICE=cuda python tools/train_step.py --batch 8 --steps 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_step.py --batch 32
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=576)
    ap.add_argument("--choice", default="s")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--no-tune", action="store_true")
    ap.add_argument("--drop-path", type=float, default=0.1, help="stochastic depth rate (reference config: 0.1)")
    args = ap.parse_args()
    os.environ.setdefault("GKG_RELPOS_DEVICE", "cuda")
    from gkgnet_amd import layers, parallel
    from gkgnet_amd.backbone import GKGNet
    from gkgnet_amd.head import LabelQueryHead, build_optimizer
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    layers.norm_cfg["type"] = "BN"                      # local batch statistics
    if not args.no_tune:                                # library GEMM selection per shape, tuned in the warm-up steps
        import torch.cuda.tunable as tunable
        tunable.enable(True); tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(10)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gkg_tunableop_train_rank{rank}.csv"))
    torch.manual_seed(0)
    net = GKGNet(choice=args.choice, n_classes=args.classes, size=args.size, drop_path=args.drop_path).to(dev).train()
    head = LabelQueryHead(args.classes, GKGNet.arch_settings[args.choice]["channels"][-1]).to(dev).train()
    parallel.broadcast_parameters(net); parallel.broadcast_parameters(head)
    params = [p for p in list(net.parameters()) + list(head.parameters()) if p.requires_grad]
    bucket = parallel.GradBucket(params)
    opt = build_optimizer([net, head])
    gen = torch.Generator().manual_seed(100 + rank)
    img = torch.randn(args.batch, 3, args.size, args.size, generator=gen).to(dev)
    tgt = (torch.rand(args.batch, args.classes, generator=gen) < 0.04).float().to(dev)

    bucket.install_overlap_hooks()       # chunk all-reduces start during the backward (the reference's DDP reducer)

    def step():
        bucket.release(prezero=True)
        feats = net(img)
        losses = head.forward_train(feats, tgt)
        loss = losses["bce_loss"] + losses["asy_loss"]
        loss.backward()
        bucket.wait()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps(dict(workload=f"GKGNet-{args.size} ({args.choice}) train step, fp32", n_gpus=world,
                              batch_per_gpu=args.batch, ms_per_step=round(1e3 * dt / args.steps, 2),
                              images_per_s=round(world * args.batch * args.steps / dt, 1), loss=float(loss),
                              peak_mem_GiB=round(torch.cuda.max_memory_allocated() / 2**30, 2))))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
