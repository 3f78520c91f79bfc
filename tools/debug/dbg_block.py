"""Where do two runs of the SAME training step differ?  (round 6)

Runs a Grapher + GrapherLabel step several times from identical seeds and compares, run against run: every projection's
tokens / pre-BN output / BN coefficients, the fp64 column sums between the projection and its apply pass, the k-NN block's
operand buffer and winning rows, and the block outputs.  Finding (profiles/r06_bn_tie_probe.txt): the sums differ in their last
bits (order of the fp64 atomics), and because every addend is rows x an fp32 tile mean the quotient S / R sits EXACTLY on an
fp32 rounding tie for about one channel in a few hundred — there the saved mean flips by one ulp between runs, 19 tokens of
the k-NN input change in one channel by one ulp and everything downstream follows.  ``fused.DETERMINISTIC`` (fixed-order sums)
does not have it.  Usage on the GPU box:  python tools/debug/dbg_block.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gkgnet_amd import block, fused, parallel
from gkgnet_amd.grapher import Grapher, GrapherLabel

def run(bucket, steps=3, with_label=True, prep=True, driver=False, dual_ok=True):
    block.ENABLED = driver
    fused.KNN_PREP = prep
    fused.DUAL_LAYOUT = dual_ok
    C, H, L, B, G, d = 64, 12, 20, 48, 2, 2
    torch.manual_seed(11)
    g = Grapher(C, 9, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True, num_group=G).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L, use_multi_group=True, num_group=G).cuda().train()
    params = list(g.parameters()) + list(gl.parameters())
    bk = parallel.GradBucket(params) if bucket else None
    gen = torch.Generator(device="cuda").manual_seed(3)
    outs = []
    for step in range(steps):
        x = torch.randn(B, C, H, H, device="cuda", generator=gen).requires_grad_(True)
        e = torch.randn(B, L, C, device="cuda", generator=gen).requires_grad_(True)
        cx, ce = torch.randn(B, C, H, H, device="cuda", generator=gen), torch.randn(B, L, C, device="cuda", generator=gen)
        if bk is not None: bk.release(prezero=True)
        else:
            for p in params: p.grad = None
        if SCR:
            from gkgnet_amd.bn_scratch import _BnBwdScratch
            sc = _BnBwdScratch.of(x.device)
            torch.cuda.synchronize()
            for i in (0, 1):
                tail = sc.store[i][sc.dirty[i]:]
                nz = tail.nonzero()
                if nz.numel():
                    print("step", step, "scratch", i, "dirty mark", sc.dirty[i], "cur", sc.cur, "nonzero beyond mark:", nz.numel(),
                          "at", (nz[:6, 0] + sc.dirty[i]).tolist(), "values", tail[nz[:6, 0]].tolist(), flush=True)
        out = g(x)
        o1 = out.detach().clone()
        torch.cuda.synchronize()
        if with_label:
            e2, edge = gl(e, out)
            torch.autograd.backward([out, e2], [cx, ce])
        else:
            out.backward(cx)
        if bk is not None: bk.pack()
        torch.cuda.synchronize()
        outs.append((o1, out.detach().clone()))
    return outs

SCR = False
SUMS = []
_chk = fused._lib.check
def _check(rc, msg=''):
    _chk(rc, msg)
    if 'statistics only' in msg:
        from gkgnet_amd.bn_scratch import _BnBwdScratch
        sc = _BnBwdScratch.of(torch.device('cuda', 0))
        torch.cuda.synchronize()
        SUMS.append((sc.cur ^ 1, sc.bufs[sc.cur ^ 1][:512].clone(), sc.bufs[sc.cur][:512].clone()))
fused._lib.check = _check
STASH = []
_real = fused._KnnMaxRelativeTM.forward
def _wrap(ctx, x, *rest):
    res = _real(ctx, x, *rest)
    XM = res[0]
    STASH.append((XM.detach(), ctx.to_save[0]))          # references only: no extra launches inside the step
    return res
fused._KnnMaxRelativeTM.forward = staticmethod(_wrap)

LIN = []
_real_lin = fused._LinearBNAct.forward
def _wrap_lin(ctx, *a, **k):
    res = _real_lin(ctx, *a, **k)
    LIN.append(tuple(ctx.to_save))
    return res
fused._LinearBNAct.forward = staticmethod(_wrap_lin)

def run2(**cfg):
    STASH.clear(); LIN.clear(); SUMS.clear()
    o = run(**cfg)
    return o, list(STASH), list(LIN), list(SUMS)

def report(tag, A, B):
    (oa, sa, la, ua), (ob, sb, lb, ub) = A, B
    for j, ((ia, ca, otha), (ib, cb, othb)) in enumerate(zip(ua, ub)):
        d = (ca != cb)
        if ia != ib or d.any() or (otha != othb).any():
            k = d.nonzero()[:4, 0].tolist()
            print(tag, 'sums call', j, 'buffer', ia, ib, 'cur differs', int(d.sum()), k, [(float(ca[i]).hex(), float(cb[i]).hex()) for i in k], 'other differs', int((otha != othb).sum()), flush=True)
    names = ("tokens", "weight", "Y", "a", "c", "mean", "invstd")
    for j, (ta, tb) in enumerate(zip(la, lb)):
        for nm_, u, v in zip(names, ta, tb):
            if u is None or v is None or u.shape != v.shape: continue
            d = (u != v)
            if d.any() and nm_ == "mean" and j == 6:
                ch = int(d.nonzero()[0, 0])
                Yd = ta[2].double()
                m = float(Yd[:, ch].mean()); fa, fb = float(u[ch]), float(v[ch])
                mid = 0.5 * (fa + fb)
                print(tag, "ch", ch, "mean_a %.17g mean_b %.17g fp64 mean %.17g  (m-mid)/ulp %.3e  |Y col| max %.3g  Y equal %s" % (
                    fa, fb, m, (m - mid) / abs(fa - fb), float(Yd[:, ch].abs().max()), bool((ta[2] == tb[2]).all())), flush=True)
                import math
                parts = Yd[:, ch].view(-1, 128).sum(1)
                print(tag, "tile sums: max |partial| %.3g, |S| %.6g" % (float(parts.abs().max()), abs(float(parts.sum()))), flush=True)
            if d.any():
                idx = d.nonzero()
                print(tag, "linear call", j, nm_, tuple(u.shape), "differs:", int(d.sum()), "first", idx[:4].tolist(),
                      "rows", sorted(set(idx[:, 0].tolist()))[:24] if idx.shape[1] > 1 else "", flush=True)
    for i, ((p, q), (r, s)) in enumerate(zip(oa, ob)):
        if (p != r).any():
            print(tag, "out differs at step", i, int((p != r).sum()), flush=True)
    for j, ((xa, aa), (xb, ab)) in enumerate(zip(sa, sb)):
        T, C2 = xa.shape
        va, vb = xa.view(T, 4, 2, C2 // 8), xb.view(T, 4, 2, C2 // 8)
        nx, nm, na = int((va[:, :, 0] != vb[:, :, 0]).sum()), int((va[:, :, 1] != vb[:, :, 1]).sum()), int((aa != ab).sum())
        if nx or nm or na:
            print(tag, "knn-mr call", j, "x differs:", nx, " m differs:", nm, " winning rows differ:", na, flush=True)

for cfg in (dict(bucket=True),):
    rs = [run2(**cfg) for _ in range(8)]
    for i in range(1, 8):
        report(f"{cfg} run{i} vs run0", rs[0], rs[i])
    print("checked", cfg, flush=True)
