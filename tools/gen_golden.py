#!/usr/bin/env python
"""Generate golden vectors for the Group-KNN hot path FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  It imports the reference's
own ``vig_model`` / ``gkgnet`` code through ``tools/ref_import.py`` and stores inputs +
expected outputs as small ``.npz`` fixtures under ``tests/golden/``.  The fixtures are
data only (tensors + ctor kwargs); no reference source is stored.

Case matrix: SURVEY.md §8(c) F1..F10.

    python tools/gen_golden.py            # regenerate everything
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_import import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# --- torch 2.10 (CPU) batch_norm-backward bug workaround -------------------------------------------------
# F.batch_norm's CPU backward returns WRONG gradients when the input and the incoming gradient have
# different memory formats (one dense NCHW, the other a channels-last strided view): checked against the
# closed-form BN gradient in float64, errors are O(10).  GrapherLabel's (B,L,C)<->(B,C,L,1) transposes
# produce exactly that mix, so un-patched golden *gradients* of the label path would pin a PyTorch bug,
# not the reference's math.  While generating we therefore run the reference with a batch_norm whose
# input and incoming gradient are made dense first (values are unchanged; forward results are identical).
class _DenseGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


_orig_batch_norm = F.batch_norm


def _dense_batch_norm(input, *args, **kwargs):
    out = _orig_batch_norm(input.contiguous(), *args, **kwargs)
    return _DenseGrad.apply(out) if out.requires_grad else out


F.batch_norm = _dense_batch_norm
torch.nn.functional.batch_norm = _dense_batch_norm


def keyed_fill_(state_dict, seed: int = 0):
    """Deterministic, construction-order-independent parameter fill: every tensor is drawn from a
    generator seeded with crc32(key) ^ seed.  The product side applies the same rule
    (tests/util.py) so no weights need to be stored for the large F10 case."""
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if key.endswith("num_batches_tracked") or key.endswith("relative_pos"):
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif ".1.weight" in key or key.endswith("bn.weight"):      # norm scale
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith(".bias"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() >= 2:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        t.copy_(v.to(t.dtype))


def randomize_norm_(module, gen):
    """Give every norm layer non-trivial affine + running statistics (seeded)."""
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def ref_top_distances(ref, knn_in_x, knn_in_y, relpos, kd):
    """Sorted top-(kd+1) distances of the matrix the reference's topk sees (for the near-tie protocol)."""
    te = ref.torch_edge
    with torch.no_grad():
        xn = F.normalize(knn_in_x, p=2.0, dim=1).transpose(2, 1).squeeze(-1)
        if knn_in_y is None:
            dist = te.pairwise_distance(xn)
        else:
            yn = F.normalize(knn_in_y, p=2.0, dim=1).transpose(2, 1).squeeze(-1)
            dist = te.xy_pairwise_distance(xn, yn)
        if relpos is not None:
            dist = dist + relpos
        kk = min(kd + 1, dist.shape[-1])
        vals, idx = torch.topk(-dist, k=kk)
    return (-vals).numpy(), idx.numpy()


def np_state(module):
    return {"sd/" + k: v.detach().cpu().clone().numpy() for k, v in module.state_dict().items()}


def save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)
    print(f"{name:32s} {os.path.getsize(path) / 1024:8.1f} KB")


# ----------------------------------------------------------------------------- Grapher cases
def grapher_case(ref, name, *, C, k, d, r, hw, G, multi, B=2, seed=0, bf16_round=False, conv="mr"):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    n = hw * hw
    mod = ref.vig.Grapher(C, k, d, conv, "gelu", "batch", True, False, 0.2, r, n=n, drop_path=0.0,
                          relative_pos=True, use_multi_group=multi, num_group=G)
    randomize_norm_(mod, gen)
    x = torch.randn(B, C, hw, hw, generator=gen)
    if bf16_round:
        x = x.to(torch.bfloat16).to(torch.float32)
    cot = torch.randn(B, C, hw, hw, generator=gen)
    sd0 = np_state(mod)                      # before any running-stat update

    cap = {}
    h1 = mod.graph_conv.register_forward_hook(lambda m, i, o: cap.update(knn_in=i[0].detach().clone(),
                                                                         edge_index=o[1].detach().clone(),
                                                                         graph=o[0].detach().clone()))
    h2 = mod.graph_conv.gconv.nn.register_forward_hook(lambda m, i, o: cap.update(nn_in=i[0].detach().clone()))
    mod.eval()
    with torch.no_grad():
        out_eval = mod(x)
    eval_edge = cap["edge_index"].clone()
    mod.train()
    xg = x.clone().requires_grad_(True)
    out = mod(xg)
    (out * cot).sum().backward()
    h1.remove(); h2.remove()

    groups = G if multi else 1
    knn_in = cap["knn_in"]                                   # (B,C,H,W) output of fc1 (train-mode BN)
    xq = knn_in.reshape(B * groups, C // groups, n, 1)
    yk = None
    if r > 1:
        yk = F.avg_pool2d(knn_in, r, r).reshape(B * groups, C // groups, -1, 1)
    topd, topi = ref_top_distances(ref, xq, yk, mod.relative_pos, k * d)
    arrays = dict(x=x.numpy(), cot=cot.numpy(), out_eval=out_eval.numpy(), out=out.detach().numpy(),
                  dx=xg.grad.numpy(), knn_in=knn_in.numpy(), edge_index=cap["edge_index"].numpy().astype(np.int32),
                  edge_index_eval=eval_edge.numpy().astype(np.int32),
                  graph=cap["graph"].numpy(), topd=topd, topi=topi.astype(np.int32))
    if conv == "mr":
        arrays["m"] = cap["nn_in"][:, 1::2, :, 0].numpy()    # (B,C,N) max-relative half of the interleave
    arrays.update(sd0)
    for pn, p in mod.named_parameters():
        if p.grad is not None:
            arrays["grad/" + pn] = p.grad.numpy()
    meta = dict(kind="grapher", C=C, k=k, dilation=d, r=r, n=n, G=G, use_multi_group=multi, B=B, hw=hw,
                conv=conv, ref="torch_vertex.py:278-333")
    save(name, meta, **arrays)


def label_case(ref, name, *, C, k, hw, L, G, multi, B=2, seed=0):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    mod = ref.vig.GrapherLabel(C, k, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=hw * hw, drop_path=0.0,
                               relative_pos=False, num_nodes=L, use_multi_group=multi, num_group=G)
    randomize_norm_(mod, gen)
    e = torch.randn(B, L, C, generator=gen)
    feat = torch.randn(B, C, hw, hw, generator=gen)
    cot = torch.randn(B, L, C, generator=gen)
    sd0 = np_state(mod)
    cap = {}
    h1 = mod.graph_conv.register_forward_hook(lambda m, i, o: cap.update(knn_in=i[0].detach().clone()))
    h2 = mod.graph_conv.gconv.nn.register_forward_hook(lambda m, i, o: cap.update(nn_in=i[0].detach().clone()))
    mod.eval()
    with torch.no_grad():
        out_eval, idx_eval = mod(e, feat)
    mod.train()
    eg = e.clone().requires_grad_(True)
    fg = feat.clone().requires_grad_(True)
    out, idx = mod(eg, fg)
    (out * cot).sum().backward()
    h1.remove(); h2.remove()
    groups = G if multi else 1
    xq = cap["knn_in"].reshape(B * groups, C // groups, L, 1)
    yk = feat.reshape(B * groups, C // groups, hw * hw, 1)
    topd, topi = ref_top_distances(ref, xq, yk, None, k)
    arrays = dict(e=e.numpy(), feat=feat.numpy(), cot=cot.numpy(), out_eval=out_eval.numpy(),
                  nn_idx_eval=idx_eval.numpy().astype(np.int32), out=out.detach().numpy(),
                  nn_idx=idx.numpy().astype(np.int32), de=eg.grad.numpy(), dfeat=fg.grad.numpy(),
                  knn_in=cap["knn_in"].numpy(), m=cap["nn_in"][:, 1::2, :, 0].numpy(), topd=topd,
                  topi=topi.astype(np.int32))
    arrays.update(sd0)
    for pn, p in mod.named_parameters():
        if p.grad is not None:
            arrays["grad/" + pn] = p.grad.numpy()
    meta = dict(kind="grapher_label", C=C, k=k, n=hw * hw, hw=hw, L=L, G=G, use_multi_group=multi, B=B,
                ref="torch_vertex.py:361-403")
    save(name, meta, **arrays)


# ----------------------------------------------------------------------------- op-level cases
def knn_op_case(ref, name, *, BG, c, N, M, k, d, relpos, seed, bf16_round=False):
    """DenseDilatedKnnGraph + MRConv2d's gather/max on raw tensors (op-level pin for a1..a7)."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(BG, c, N, 1, generator=gen)
    y = None if M is None else torch.randn(BG, c, M, 1, generator=gen)
    if bf16_round:
        x = x.to(torch.bfloat16).float()
        y = None if y is None else y.to(torch.bfloat16).float()
    rp = None
    if relpos:
        rp = -torch.rand(1, N, N if M is None else M, generator=gen)
    g = ref.torch_edge.DenseDilatedKnnGraph(k, d, False, 0.0)
    edge = g(x, y, rp)
    bis = ref.torch_nn.batched_index_select
    x_i = bis(x, edge[1])
    x_j = bis(x if y is None else y, edge[0])
    rel, arg = torch.max(x_j - x_i, -1)
    gcot = torch.randn(BG, c, N, generator=gen)
    xg = x.clone().requires_grad_(True)
    yg = None if y is None else y.clone().requires_grad_(True)
    xi = bis(xg, edge[1]); xj = bis(xg if yg is None else yg, edge[0])
    mm, _ = torch.max(xj - xi, -1)
    (mm * gcot).sum().backward()
    topd, topi = ref_top_distances(ref, x, y, rp, k * d)
    arrays = dict(x=x.squeeze(-1).numpy(), edge_index=edge.numpy().astype(np.int32), m=rel.numpy(),
                  gcot=gcot.numpy(), dx=xg.grad.squeeze(-1).numpy(), topd=topd, topi=topi.astype(np.int32))
    if y is not None:
        arrays["y"] = y.squeeze(-1).numpy()
        arrays["dy"] = yg.grad.squeeze(-1).numpy()
    if rp is not None:
        arrays["relpos"] = rp.numpy()
    meta = dict(kind="knn_op", BG=BG, c=c, N=N, M=M, k=k, dilation=d, bf16_round=bf16_round,
                ref="torch_edge.py:164-176; torch_vertex.py:49-54")
    save(name, meta, **arrays)


def integer_kat(ref, name):
    """F8: small-integer features straight into the UN-normalised op-level functions
    (dense_knn_matrix / xy_dense_knn_matrix, torch_edge.py:54,89) where fp32 math is exact.
    Tie-free by rejection sampling: all squared distances are integers < 2^24 and the sorted
    top-(k+1) of every query is strictly increasing."""
    te = ref.torch_edge
    c, N, M, k = 6, 24, 40, 7

    def tie_free(dist):
        srt = torch.sort(dist, dim=-1).values
        return bool((srt[..., 1:k + 1] - srt[..., :k]).min() >= 1.0)

    for seed in range(1000):                       # rejection-sample a tie-free instance
        rng = np.random.RandomState(seed)
        x = rng.randint(-30, 31, size=(2, c, N, 1)).astype(np.float32)
        y = rng.randint(-30, 31, size=(2, c, M, 1)).astype(np.float32)
        xt, yt = torch.from_numpy(x), torch.from_numpy(y)
        dist = te.xy_pairwise_distance(xt.transpose(2, 1).squeeze(-1), yt.transpose(2, 1).squeeze(-1))
        dself = te.pairwise_distance(yt.transpose(2, 1).squeeze(-1))
        if tie_free(dist) and tie_free(dself):
            break
    else:
        raise RuntimeError("no tie-free KAT instance found")
    e_xy = te.xy_dense_knn_matrix(xt, yt, k, None)
    e_self = te.dense_knn_matrix(yt, k, None)
    meta = dict(kind="integer_kat", c=c, N=N, M=M, k=k, ref="torch_edge.py:54-106 (no normalisation)")
    save(name, meta, x=x[..., 0], y=y[..., 0], edge_xy=e_xy.numpy().astype(np.int32),
         edge_self=e_self.numpy().astype(np.int32), dist_xy=dist.numpy(), dist_self=dself.numpy())


def relpos_case(ref, name, combos):
    arrays = {}
    metas = []
    for (C, n, r) in combos:
        mod = ref.vig.Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, r, n=n, relative_pos=True)
        arrays[f"rp_{C}_{n}_{r}"] = mod.relative_pos.detach().numpy()
        metas.append([C, n, r])
    # runtime re-interpolation path (torch_vertex.py:317-323): built for n=64, run at 10x10
    mod = ref.vig.Grapher(32, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 2, n=64, relative_pos=True)
    arrays["rp_runtime_32_64_2_to_10x10"] = mod._get_relative_pos(mod.relative_pos, 10, 10).detach().numpy()
    save(name, dict(kind="relpos", combos=metas, ref="pos_embed.py:21-85; torch_vertex.py:309-323"), **arrays)


def backbone_case(ref, name):
    """F10: tiny full GKGNet forward (wiring pin).  Weights are NOT stored: both sides fill the
    state_dict with keyed_fill_ (crc32(key)-seeded)."""
    torch.manual_seed(0)
    kw = dict(choice="t", k=4, k_label_gcn=4, n_classes=8, size=128, drop_path=0.0)
    net = ref.gkgnet.GKGNet(**kw)
    sd = net.state_dict()
    with torch.no_grad():
        keyed_fill_(sd, seed=10)
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(123)
    img = torch.randn(1, 3, 128, 128, generator=gen)
    net.eval()
    # per-block intermediates: a randomly initialised 16-layer k-NN network is chaotic (one flipped near-tie
    # neighbour perturbs everything downstream), so parity is pinned block by block on the reference's own inputs
    cap = {}
    hooks = [net.stem.register_forward_hook(lambda m, i, o: cap.__setitem__("stem", o.detach().clone()))]
    for bi, blk in enumerate(net.backbone):
        hooks.append(blk.register_forward_hook(lambda m, i, o, bi=bi: cap.__setitem__(f"x{bi}", o.detach().clone())))
    for si in range(4):
        for li, gl in enumerate(net.gcn_label[si]):
            hooks.append(gl.register_forward_hook(
                lambda m, i, o, si=si, li=li: cap.update({f"lab_in{si}_{li}": i[0].detach().clone(),
                                                          f"lab_out{si}_{li}": o[0].detach().clone(),
                                                          f"lab_edge{si}_{li}": o[1].detach().clone()})))
    with torch.no_grad():
        e, gap, edge = net(img)
    for h in hooks:
        h.remove()
    keys = sorted(sd.keys())
    shapes = {k: list(sd[k].shape) for k in keys}
    meta = dict(kind="backbone", ctor=kw, state_shapes=shapes, ref="gkgnet.py:150-284")
    arrays = {k: v.numpy() if v.dtype != torch.int64 else v.numpy().astype(np.int32) for k, v in cap.items()}
    save(name, meta, img=img.numpy(), label_tokens=e.numpy(), gap=gap.numpy(), edge_index=edge.numpy().astype(np.int32),
         **arrays)


def head_case(ref, name):
    """F12: LabelQueryHead scores + double loss (ASL x10 + smoothed multi-label BCE) and gradients.
    Config as in configs/gkgnet/gkgnet_coco_576.py:27-38 at a small size."""
    torch.manual_seed(12)
    L, C, B = 8, 32, 5
    head = ref.head.LabelQueryHead(num_classes=L, in_channels=C, softmax=False,
                                   loss=dict(type="AsymmetricLoss", gamma_pos=0.0, gamma_neg=2.0, clip=0.05), topk=(1, 1))
    gen = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for p in head.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
    e = (torch.randn(B, L, C, generator=gen)).requires_grad_(True)
    gap = (torch.randn(B, C, generator=gen)).requires_grad_(True)
    gt = (torch.rand(B, L, generator=gen) < 0.3).float()
    losses = head.forward_train((e, gap, None), gt)
    total = losses["bce_loss"] + losses["asy_loss"]
    total.backward()
    with torch.no_grad():
        score = head.get_score((e, gap))
        pred = head.simple_test((e, gap, None), post_process=False)
    arrays = dict(e=e.detach().numpy(), gap=gap.detach().numpy(), gt=gt.numpy(), score=score.numpy(), pred=pred.numpy(),
                  bce_loss=losses["bce_loss"].detach().numpy(), asy_loss=losses["asy_loss"].detach().numpy(),
                  de=e.grad.numpy(), dgap=gap.grad.numpy())
    for k, v in head.state_dict().items():
        arrays["sd/" + k] = v.detach().clone().numpy()
    for k, p in head.named_parameters():
        arrays["grad/" + k] = p.grad.numpy()
    save(name, dict(kind="head", num_classes=L, in_channels=C, B=B, ref="heads/label_query_head.py:10-85; "
                    "losses/asymmetric_loss.py:9-71; losses/label_smooth_loss.py:122-175"), **arrays)


# ----------------------------------------------------------------------------- autocast (bf16) cases
def autocast_case(ref, name, *, C, k, d, hw, G, L, B=2, seed=0):
    """F14: the reference's Grapher -> GrapherLabel chain evaluated under torch.autocast('cpu', bfloat16) (eval mode,
    no_grad: BASELINE configs 3/5 are bf16 inference) next to the same chain in fp32.  Under autocast the reference runs
    its 1x1 convolutions AND the pairwise-distance matmul in bf16, so its own graph differs from the fp32 one; the
    fixture stores both runs so that the product's mixed-precision path (bf16 GEMM operands, fp32 accumulation, fp32
    k-NN) is held to: at least as close to the fp32 reference as the reference's own autocast run is."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    n = hw * hw
    g = ref.vig.Grapher(C, k, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=n, drop_path=0.0, relative_pos=True,
                        use_multi_group=True, num_group=G)
    gl = ref.vig.GrapherLabel(C, k, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=n, drop_path=0.0,
                              relative_pos=False, num_nodes=L, use_multi_group=True, num_group=G)
    randomize_norm_(g, gen)
    randomize_norm_(gl, gen)
    x = torch.randn(B, C, hw, hw, generator=gen)
    e = torch.randn(B, L, C, generator=gen)
    g.eval(); gl.eval()
    cap = {}
    h = g.graph_conv.register_forward_hook(lambda m, i, o: cap.update(edge=o[1].detach().clone()))

    def run():
        out = g(x)
        e2, idx = gl(e, out.float())
        return out.float(), e2.float(), cap["edge"].clone(), idx.clone()

    with torch.no_grad():
        out32, e32, edge32, idx32 = run()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out_ac, e_ac, edge_ac, idx_ac = run()
    h.remove()
    arrays = dict(x=x.numpy(), e=e.numpy(), out_fp32=out32.numpy(), labels_fp32=e32.numpy(),
                  edge_fp32=edge32.numpy().astype(np.int32), idx_fp32=idx32.numpy().astype(np.int32),
                  out_autocast=out_ac.numpy(), labels_autocast=e_ac.numpy(),
                  edge_autocast=edge_ac.numpy().astype(np.int32), idx_autocast=idx_ac.numpy().astype(np.int32))
    arrays.update({"g/" + k_: v for k_, v in np_state(g).items()})
    arrays.update({"gl/" + k_: v for k_, v in np_state(gl).items()})
    meta = dict(kind="autocast", C=C, k=k, dilation=d, hw=hw, n=n, G=G, L=L, B=B,
                ref="torch_vertex.py:278-403 under torch.autocast('cpu', torch.bfloat16), eval, no_grad")
    save(name, meta, **arrays)


def amp_fp16_case(ref, name, *, C, k, d, hw, G, L, B=2, seed=0, loss_scale=128.0):
    """F17: the reference's own training recipe is fp16 AMP (configs/gkgnet/gkgnet_coco_576.py:146 fp16 = dict(loss_scale=
    'dynamic'); mmcls/models/classifiers/base.py:80 auto_fp16; mmcls/core/fp16/hooks.py:13-129 — with torch >= 1.6 mmcv
    runs the forward under torch.cuda.amp.autocast and scales the loss).  The reference's Grapher -> GrapherLabel chain in
    TRAIN mode, forward + backward, (i) in fp32 and (ii) under torch.autocast('cpu', float16) with the loss multiplied by
    ``loss_scale`` and the gradients divided by it afterwards (CPU half matmul / conv are usable in this torch build: the
    autocast definition itself, not an fp16-rounded-inputs stand-in).  The product's fp16-autocast path is held to the F14
    rule: at least as close to the fp32 reference as the reference's own fp16 run is (outputs, input and parameter
    gradients)."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    n = hw * hw
    g = ref.vig.Grapher(C, k, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=n, drop_path=0.0, relative_pos=True,
                        use_multi_group=True, num_group=G)
    gl = ref.vig.GrapherLabel(C, k, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=n, drop_path=0.0,
                              relative_pos=False, num_nodes=L, use_multi_group=True, num_group=G)
    randomize_norm_(g, gen)
    randomize_norm_(gl, gen)
    x0 = torch.randn(B, C, hw, hw, generator=gen)
    e0 = torch.randn(B, L, C, generator=gen)
    cot_x = torch.randn(B, C, hw, hw, generator=gen) / (C * n) ** 0.5
    cot_e = torch.randn(B, L, C, generator=gen) / (C * L) ** 0.5
    sd_g = {k_: v.clone() for k_, v in g.state_dict().items()}
    sd_l = {k_: v.clone() for k_, v in gl.state_dict().items()}
    watch_g = ["fc1.0.weight", "graph_conv.gconv.nn.0.weight", "fc2.0.weight", "fc2.1.weight", "fc2.1.bias"]
    watch_l = ["fc1.0.weight", "graph_conv.gconv.nn.0.weight", "ffn.fc1.0.weight", "ffn.fc2.0.weight", "ffn.fc2.1.weight"]

    def run(amp):
        g.load_state_dict(sd_g); gl.load_state_dict(sd_l)           # running statistics restart for each leg
        g.train(); gl.train()
        g.zero_grad(set_to_none=True); gl.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        e = e0.clone().requires_grad_(True)
        if amp:
            with torch.autocast("cpu", dtype=torch.float16):
                out = g(x)
                e2, idx = gl(e, out)
            loss = ((out.float() * cot_x).sum() + (e2.float() * cot_e).sum()) * loss_scale
        else:
            out = g(x)
            e2, idx = gl(e, out)
            loss = (out * cot_x).sum() + (e2 * cot_e).sum()
        loss.backward()
        inv = 1.0 / loss_scale if amp else 1.0
        res = dict(out=out.detach().float().numpy(), labels=e2.detach().float().numpy(), idx=idx.numpy().astype(np.int32),
                   dx=(x.grad * inv).numpy(), de=(e.grad * inv).numpy())
        pg, pl = dict(g.named_parameters()), dict(gl.named_parameters())
        for w in watch_g:
            res["g/grad/" + w] = (pg[w].grad.float() * inv).numpy()
        for w in watch_l:
            res["gl/grad/" + w] = (pl[w].grad.float() * inv).numpy()
        return res

    r32 = run(False)
    r16 = run(True)
    arrays = dict(x=x0.numpy(), e=e0.numpy(), cot_x=cot_x.numpy(), cot_e=cot_e.numpy())
    arrays.update({"fp32/" + k_: v for k_, v in r32.items()})
    arrays.update({"amp/" + k_: v for k_, v in r16.items()})
    arrays.update({"g/sd/" + k_: v.numpy() for k_, v in sd_g.items()})
    arrays.update({"gl/sd/" + k_: v.numpy() for k_, v in sd_l.items()})
    finite = all(np.isfinite(v).all() for k_, v in r16.items() if v.dtype.kind == "f")
    meta = dict(kind="amp_fp16", C=C, k=k, dilation=d, hw=hw, n=n, G=G, L=L, B=B, loss_scale=loss_scale, finite=bool(finite),
                watch_g=watch_g, watch_l=watch_l,
                ref="torch_vertex.py:278-403 train mode fwd+bwd under torch.autocast('cpu', torch.float16), loss x loss_scale; "
                    "configs/gkgnet/gkgnet_coco_576.py:146; mmcls/core/fp16/hooks.py:13-129")
    save(name, meta, **arrays)
    for k_ in ("out", "labels", "dx", "de"):
        print(f"  F17 {k_}: mean|amp - fp32| = {np.abs(r16[k_] - r32[k_]).mean():.3e}  (scale {np.abs(r32[k_]).mean():.3e})")


# ----------------------------------------------------------------------------- train-step case
def paramwise_groups(modules, weight_decay):
    """The reference's paramwise_cfg (configs/gkgnet/gkgnet_coco_576.py:110-117: norm_decay_mult=0, bias_decay_mult=0)
    as mmcv's DefaultOptimizerConstructor applies it: every parameter of a norm layer and every 'bias' decays with 0."""
    decay, no_decay = [], []
    for mod in modules:
        for m in mod.modules():
            is_norm = isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.LayerNorm, torch.nn.GroupNorm))
            for pn, p in m.named_parameters(recurse=False):
                if p.requires_grad:
                    (no_decay if (is_norm or pn == "bias") else decay).append(p)
    return [dict(params=decay, weight_decay=weight_decay), dict(params=no_decay, weight_decay=0.0)]


def train_step_case(ref, name, steps=2):
    """F15 (SURVEY §8 f2 / BASELINE config 4 at a tiny size): the reference's GKGNet backbone + LabelQueryHead in train
    mode, loss = smoothed BCE + 10 x ASL (heads/label_query_head.py:70-85), AdamW lr 1e-4 wd 0.05 with the paramwise
    config, grad-clip 5.0 (gkgnet_coco_576.py:110-126).  Stores per-step loss / pre-clip gradient norm and a few
    parameters after `steps` updates.  Weights: keyed_fill_ (both sides regenerate them)."""
    torch.manual_seed(0)
    kw = dict(choice="t", k=4, k_label_gcn=4, n_classes=8, size=128, drop_path=0.0)
    net = ref.gkgnet.GKGNet(**kw)
    sd = net.state_dict()
    with torch.no_grad():
        keyed_fill_(sd, seed=15)
    net.load_state_dict(sd)
    head = ref.head.LabelQueryHead(num_classes=8, in_channels=384, softmax=False,
                                   loss=dict(type="AsymmetricLoss", gamma_pos=0.0, gamma_neg=2.0, clip=0.05), topk=(1, 1))
    hsd = head.state_dict()
    with torch.no_grad():
        keyed_fill_(hsd, seed=16)
    head.load_state_dict(hsd)
    net.train(); head.train()
    gen = torch.Generator().manual_seed(150)
    B = 4
    img = torch.randn(B, 3, 128, 128, generator=gen).to(torch.bfloat16).float()     # bf16-representable: stored as 16 bits
    gt = (torch.rand(B, 8, generator=gen) < 0.3).float()
    params = [p for p in list(net.parameters()) + list(head.parameters()) if p.requires_grad]
    opt = torch.optim.AdamW(paramwise_groups([net, head], 0.05), lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    watch = ["backbone.0.0.fc1.0.weight", "backbone.0.0.graph_conv.gconv.nn.0.weight", "backbone.7.0.fc2.0.weight",
             "backbone.13.0.fc1.1.weight", "gcn_label.0.0.ffn.fc1.0.weight", "gcn_label.3.0.graph_conv.gconv.nn.1.bias",
             "label_lt.weight", "backbone.4.0.fc2.1.weight", "pos_embed"]
    named = dict(net.named_parameters())
    before = {k_: named[k_].detach().clone() for k_ in watch}
    hbefore = head.fc1.weight.detach().clone()
    losses, bces, asys, norms = [], [], [], []
    grads0 = {}
    # the graphs the reference builds, in call order (16 DenseDilatedKnnGraph modules per forward): lets the product
    # be run ON THE SAME GRAPHS, which removes the only chaotic element (fp32 near-tie neighbour flips) from the comparison
    graphs = []
    hooks = [m.register_forward_hook(lambda mod, i, o: graphs.append(o.detach()[0].clone()))
             for m in net.modules() if isinstance(m, ref.torch_edge.DenseDilatedKnnGraph)]
    for it in range(steps):
        opt.zero_grad(set_to_none=True)
        out = head.forward_train(net(img), gt)
        loss = out["bce_loss"] + out["asy_loss"]
        loss.backward()
        if it == 0:
            grads0 = {k_: named[k_].grad.detach().clone() for k_ in watch}
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 5.0)))
        opt.step()
        losses.append(float(loss)); bces.append(float(out["bce_loss"])); asys.append(float(out["asy_loss"]))
    for h_ in hooks:
        h_.remove()
    assert len(graphs) == 16 * steps
    arrays = dict(img_bf16=img.to(torch.bfloat16).view(torch.int16).numpy(), gt=gt.numpy(), loss=np.array(losses), bce_loss=np.array(bces),
                  asy_loss=np.array(asys), grad_norm=np.array(norms), head_fc1_delta=(head.fc1.weight.detach() - hbefore).numpy())
    for gi, gr in enumerate(graphs):
        arrays[f"graph/{gi:02d}"] = gr.numpy().astype(np.int16)
    for k_ in watch:
        arrays["delta/" + k_] = (named[k_].detach() - before[k_]).numpy()
        arrays["grad0/" + k_] = grads0[k_].numpy()
    meta = dict(kind="train_step", ctor=kw, head=dict(num_classes=8, in_channels=384), B=B, steps=steps, lr=1e-4,
                weight_decay=0.05, grad_clip=5.0, watch=watch,
                ref="gkgnet.py:150-284; heads/label_query_head.py:70-85; configs/gkgnet/gkgnet_coco_576.py:110-126")
    save(name, meta, **arrays)


def main():
    ref = load_reference(with_backbone=True, with_head=True)
    only = set(sys.argv[1:])
    if only:
        if 'f14' in only:
            autocast_case(ref, 'f14_autocast_bf16', C=64, k=9, d=2, hw=12, G=2, L=20, seed=14)
        if 'f15' in only:
            train_step_case(ref, 'f15_train_step')
        if 'f17' in only:
            amp_fp16_case(ref, 'f17_amp_fp16', C=64, k=9, d=2, hw=12, G=2, L=20, seed=17)
        return
    grapher_case(ref, "f1_grapher_cfg1", C=64, k=9, d=1, r=1, hw=14, G=1, multi=False)
    grapher_case(ref, "f2_grapher_g4", C=64, k=9, d=1, r=1, hw=8, G=4, multi=True, seed=2)
    grapher_case(ref, "f3_grapher_dil3", C=64, k=9, d=3, r=1, hw=12, G=2, multi=True, seed=3)
    grapher_case(ref, "f4a_grapher_r2", C=32, k=9, d=1, r=2, hw=16, G=2, multi=True, seed=4)
    grapher_case(ref, "f4b_grapher_r4", C=32, k=9, d=1, r=4, hw=16, G=2, multi=True, seed=5)
    grapher_case(ref, "f7_grapher_bf16in", C=64, k=9, d=2, r=1, hw=8, G=2, multi=True, seed=7, bf16_round=True)
    grapher_case(ref, "f11_grapher_edgeconv", C=32, k=6, d=1, r=1, hw=6, G=1, multi=False, seed=11, conv="edge")
    label_case(ref, "f5_label_g2", C=64, k=9, hw=8, L=80, G=2, multi=True, seed=5)
    label_case(ref, "f5b_label_g1", C=32, k=5, hw=6, L=12, G=1, multi=False, seed=6)
    knn_op_case(ref, "op_self_relpos", BG=3, c=20, N=70, M=None, k=5, d=2, relpos=True, seed=21)
    knn_op_case(ref, "op_xy_norelpos", BG=4, c=12, N=33, M=150, k=9, d=1, relpos=False, seed=22)
    knn_op_case(ref, "op_xy_relpos_dil", BG=2, c=48, N=100, M=25, k=4, d=3, relpos=True, seed=23)
    knn_op_case(ref, "op_self_bf16", BG=2, c=40, N=96, M=None, k=9, d=1, relpos=False, seed=24, bf16_round=True)
    knn_op_case(ref, "op_label_like", BG=2, c=32, N=80, M=400, k=9, d=1, relpos=False, seed=25)
    integer_kat(ref, "f8_integer_kat")
    relpos_case(ref, "f9_relpos", [(64, 196, 1), (80, 144, 1), (32, 256, 2), (32, 256, 4)])
    backbone_case(ref, "f10_backbone_tiny")
    head_case(ref, "f12_head_loss")
    autocast_case(ref, 'f14_autocast_bf16', C=64, k=9, d=2, hw=12, G=2, L=20, seed=14)
    train_step_case(ref, 'f15_train_step')
    amp_fp16_case(ref, 'f17_amp_fp16', C=64, k=9, d=2, hw=12, G=2, L=20, seed=17)
    coco_case('f16_coco')
    map_case("f13_map")



def map_case(name="f13_map"):
    """F13: mAP of random scores vs the reference's mean_ap.py (numpy-only file, imported by path)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_mean_ap", os.path.join(
        os.environ.get("GKG_REFERENCE_ROOT", "/root/reference"), "mmcls/core/evaluation/mean_ap.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.RandomState(4)
    pred = rng.rand(200, 12).astype(np.float32)
    target = (rng.rand(200, 12) < 0.15).astype(np.int64)
    target[rng.rand(200, 12) < 0.03] = -1
    target[:, 5] = 0
    target[3, 5] = 1
    aps = [m.average_precision(pred[:, k], target[:, k]) for k in range(12)]
    save(name, dict(kind="map", ref="core/evaluation/mean_ap.py:6-74"), pred=pred, target=target,
         mAP=np.float64(m.mAP(pred, target)), ap=np.array(aps))

def coco_case(name="f16_coco"):
    """F16: the reference's COCO dataset class — its pickled-annotation reader and get_coco_metrics —
    run on a synthetic annotation file and random scores.  mmcls/datasets/coco.py is imported by path with stand-ins for
    the packages its module header pulls in (mmcv, the mmcls dataset registry / base class, pipelines); scikit-learn,
    which get_coco_metrics calls, is the real one."""
    import importlib.util
    import pickle
    import tempfile
    import types
    root = os.environ.get("GKG_REFERENCE_ROOT", "/root/reference")

    def mod(name_, **attrs):
        m = types.ModuleType(name_)
        m.__dict__.update(attrs)
        sys.modules[name_] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    class _Base:                                              # stands in for MultiLabelDataset (abstract plumbing only)
        def get_gt_labels(self):
            return np.array([d["gt_label"] for d in self.data_infos])
    saved = {k_: sys.modules.get(k_) for k_ in ("mmcv", "mmcls", "mmcls.datasets", "mmcls.datasets.builder",
                                                "mmcls.datasets.multi_label", "mmcls.core", "mmcls.core.evaluation",
                                                "mmcls.models", "mmcls.models.losses", "mmcls.datasets.pipelines", "easydict")}
    mod("mmcv")
    for pk in ("mmcls", "mmcls.datasets", "mmcls.core", "mmcls.models"):
        m = mod(pk); m.__path__ = []
    mod("mmcls.datasets.builder", DATASETS=_Reg())
    mod("mmcls.datasets.multi_label", MultiLabelDataset=_Base)
    mod("mmcls.core.evaluation", precision_recall_f1=None, support=None)
    mod("mmcls.models.losses", accuracy=None)
    mod("mmcls.datasets.pipelines", Compose=None)
    mod("easydict", EasyDict=dict)
    spec = importlib.util.spec_from_file_location("mmcls.datasets.coco", os.path.join(root, "mmcls/datasets/coco.py"))
    ref_coco = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_coco)
    rng = np.random.RandomState(16)
    N, C = 60, 80
    records = [dict(file_name=f"COCO_val2014_{i:012d}.jpg", objects=(rng.rand(C) < 0.06).astype(np.float32), area=None)
               for i in range(N)]
    for r in records[:3]:
        r["objects"][0] = 1.0
    with tempfile.TemporaryDirectory() as td:
        ann = os.path.join(td, "val_test.data")
        with open(ann, "wb") as fh:
            pickle.dump(records, fh)
        ds = object.__new__(ref_coco.COCO)
        ds.ann_file, ds.data_prefix = ann, "../0data/coco/val2014"
        ds.data_infos = ds.load_annotations()
        ann_bytes = np.frombuffer(open(ann, "rb").read(), dtype=np.uint8)
    gt = ds.get_gt_labels()
    preds = (rng.rand(N, C) ** 3).astype(np.float32)
    preds[gt == 1] = np.clip(preds[gt == 1] + 0.45, 0, 1)         # an informative classifier: positives score higher
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        met = ref_coco.get_coco_metrics(gt, preds, threshold=0.5)
    for k_, v in saved.items():
        if v is None:
            sys.modules.pop(k_, None)
        else:
            sys.modules[k_] = v
    meta = dict(kind="coco", N=N, C=C, data_prefix=ds.data_prefix, filenames=[d["img_info"]["filename"] for d in ds.data_infos],
                metrics={k_: float(v) for k_, v in met.items()}, ref="mmcls/datasets/coco.py:65-176,261-285")
    save(name, meta, ann_file_bytes=ann_bytes, gt=gt.astype(np.int8), preds=preds)


if __name__ == "__main__":
    if "f16" in sys.argv[1:]:
        coco_case()
    else:
        main()
