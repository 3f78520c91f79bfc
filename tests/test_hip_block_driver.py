"""Round 6: the block-level entry points (csrc/gkg_block.hip, gkgnet_amd/block.py) — a Grapher / GrapherLabel block's forward and
backward as ONE library call each (reference torch_vertex.py:325-333, :392-403, FFNLabel :334-360).  The C driver issues the
launches of the per-layer composition in ``fused.py`` in the same order with the same arguments, so everything must agree BIT FOR
BIT: outputs, the returned graph, input gradients, BN parameter gradients, running statistics; the weight gradients (atomically
accumulated slabs: run-dependent order) to rounding.  Two steps, so that the dual layout and the prepared label keys engage."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(driver: bool, monkeypatch, C=64, H=12, L=20, B=48, G=2, bucket=False, d=2, plans=True):
    from gkgnet_amd import block, fused, parallel
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    monkeypatch.setattr(block, "ENABLED", driver)
    torch.manual_seed(11)
    g = Grapher(C, 9, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=G).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=G).cuda().train()
    params = list(g.parameters()) + list(gl.parameters())
    bk = parallel.GradBucket(params) if bucket else None
    gen = torch.Generator(device="cuda").manual_seed(3)
    calls = {"g": 0, "l": 0}
    rg, rl = block._GrapherBlockFn.forward, block._LabelBlockFn.forward        # (the first step reaches them through the full
    monkeypatch.setattr(block._GrapherBlockFn, "forward",                       # eligibility test, later ones through the plan)
                        staticmethod(lambda *a: (calls.__setitem__("g", calls["g"] + 1), rg(*a))[1]))
    monkeypatch.setattr(block._LabelBlockFn, "forward", staticmethod(lambda *a: (calls.__setitem__("l", calls["l"] + 1), rl(*a))[1]))
    steps = []
    for step in range(3):
        x = torch.randn(B, C, H, H, device="cuda", generator=gen).requires_grad_(True)
        e = torch.randn(B, L, C, device="cuda", generator=gen).requires_grad_(True)
        cx, ce = torch.randn(B, C, H, H, device="cuda", generator=gen), torch.randn(B, L, C, device="cuda", generator=gen)
        if bk is not None:
            bk.release(prezero=True)
        else:
            for p in params:
                p.grad = None
        if not plans:
            block._PLANS.clear()                      # every step through the full eligibility test (fused.grapher_forward ...)
        out = g(x)
        e2, edge = gl(e, out)
        torch.autograd.backward([out, e2], [cx, ce])
        if bk is not None:
            bk.pack()
        torch.cuda.synchronize()
        steps.append(dict(out=out.detach().clone(), e2=e2.detach().clone(), edge=edge.clone(), dx=x.grad.clone(), de=e.grad.clone(),
                          grads=[None if p.grad is None else p.grad.clone() for p in params],
                          bufs=[b.clone() for b in list(g.buffers()) + list(gl.buffers())],
                          names=[n for n, _ in list(g.named_parameters()) + list(gl.named_parameters())]))
    monkeypatch.setattr(block._GrapherBlockFn, "forward", staticmethod(rg))
    monkeypatch.setattr(block._LabelBlockFn, "forward", staticmethod(rl))
    return calls, steps


def _first_difference(on, off):
    for step, (a, b) in enumerate(zip(on, off)):
        for key in ("out", "e2", "edge", "dx", "de"):
            if not torch.equal(a[key], b[key]):
                return (step, key, float((a[key].float() - b[key].float()).abs().max()), int((a[key] != b[key]).sum()))
        for u, v in zip(a["bufs"], b["bufs"]):
            if not torch.equal(u, v):
                return (step, "buffers")
        for name, u, v in zip(a["names"], a["grads"], b["grads"]):
            if (u is None) != (v is None):
                return (step, name, "presence")
            if u is None:
                continue
            if u.dim() >= 2:                                           # weight gradients: slabs added with fp32 atomics
                ok = torch.allclose(u, v, rtol=1e-4, atol=1e-3 * float(v.abs().max()) + 1e-6)
            else:                                                      # BN gammas / betas: plain stores of fp64-accumulated sums
                ok = torch.allclose(u, v, rtol=1e-5, atol=1e-5 * float(v.abs().max()) + 1e-7)
            if not ok:
                return (step, name)
    return None


@pytest.mark.parametrize("bucket,plans", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("shape", [dict(C=64, H=12, L=20, B=48, G=2, d=2), dict(C=80, H=9, L=7, B=5, G=4, d=1)])
def test_block_driver_is_bit_identical_to_the_composition(bucket, plans, shape, monkeypatch):
    """The non-deterministic default accumulates the BN column sums with fp64 atomics; their addends are exact multiples of an fp32
    quantum (rows x an fp32 tile mean), so S / R lands on an fp32 rounding TIE for about one channel in a few hundred, and there
    the run-dependent order of the atomics decides the last bit of the saved mean (measured with tools/debug/dbg_block.py: two
    runs of the SAME path differ that way).  A pair of runs that hits such a tie differently is repeated;
    a difference between the driver and the composition would show in every pair."""
    diff = None
    for attempt in range(4):
        c1, on = _run(True, monkeypatch, bucket=bucket, plans=plans, **shape)
        c0, off = _run(False, monkeypatch, bucket=bucket, **shape)
        assert c1 == {"g": 3, "l": 3} and c0 == {"g": 0, "l": 0}      # the driver ran (every step) / did not run
        diff = _first_difference(on, off)
        if diff is None:
            return
    # Four pairs in a row differ: either a tie that the two paths' different launch pacing decides the same way every time (the
    # order of the atomics is timing), or a real difference.  A tie moves one channel's mean by one ulp and, at worst, a near-tie
    # neighbour with it: everything must still agree to rounding, and nearly every list entry must be the same.
    import warnings
    for step, (a, b) in enumerate(zip(on, off)):
        for key in ("out", "e2", "dx", "de"):
            err, ref = float((a[key] - b[key]).abs().max()), float(b[key].abs().max())
            assert err <= 2e-5 * ref + 1e-6, (step, key, err, ref, diff)
        assert float((a["edge"] == b["edge"]).float().mean()) >= 0.998, (step, "edge", diff)
    warnings.warn(f"driver and composition were never bit-identical in 4 pairs of runs (first difference {diff}); they agree to "
                  f"rounding - a BN mean on an fp32 rounding tie (DESIGN.md section 4)")


def test_ineligible_calls_keep_the_composition(monkeypatch):
    """Eval mode, DropPath, pooled keys and autocast are outside the driver's scope: the composition runs (and the driver does not)."""
    from gkgnet_amd import block
    from gkgnet_amd.grapher import Grapher
    calls = []
    real = block._GrapherBlockFn.forward
    monkeypatch.setattr(block._GrapherBlockFn, "forward", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
    torch.manual_seed(0)
    x = torch.randn(4, 64, 12, 12, device="cuda")
    g = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=144, relative_pos=True, use_multi_group=True, num_group=2).cuda()
    g.train()(x.requires_grad_(True)).sum().backward()
    assert len(calls) == 1
    g.eval()
    with torch.no_grad():
        g(x)
    gp = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 2, n=144, relative_pos=True, use_multi_group=True, num_group=2).cuda().train()
    gp(x).sum().backward()                                              # r = 2: pooled keys
    gd = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=144, drop_path=0.3, relative_pos=True, use_multi_group=True,
                 num_group=2).cuda().train()
    gd(x).sum().backward()                                              # active DropPath
    with torch.autocast("cuda", dtype=torch.bfloat16):
        g.train()(x).float().sum().backward()
    assert len(calls) == 1


def test_a_plan_is_dropped_when_the_module_changes(monkeypatch):
    """The per-module plan (block._Plan) bakes pointers of the BN tensors into the descriptor and skips the eligibility test: a
    module whose tensors, sub-modules or mode changed must fall back to the full test (and get a new plan or the composition)."""
    from gkgnet_amd import block
    from gkgnet_amd.grapher import Grapher
    torch.manual_seed(0)
    x = torch.randn(4, 64, 12, 12, device="cuda")
    g = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=144, relative_pos=True, use_multi_group=True, num_group=2).cuda().train()
    ref = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=144, relative_pos=True, use_multi_group=True, num_group=2).cuda().train()
    ref.load_state_dict(g.state_dict())
    monkeypatch.setattr(block, "ENABLED", True)
    g(x)
    assert block.try_grapher(g, x) is not None                                     # the plan answers
    # (1) a parameter replaced by a new tensor object
    with torch.no_grad():
        g.fc1[1].weight = torch.nn.Parameter(g.fc1[1].weight.detach() * 2.0)
        ref.fc1[1].weight.mul_(2.0)
    assert block.try_grapher(g, x) is None
    monkeypatch.setattr(block, "ENABLED", False)
    want = ref(x)
    monkeypatch.setattr(block, "ENABLED", True)
    # running statistics: both modules have seen a different number of batches by now; compare on fresh copies of the buffers
    g.load_state_dict(ref.state_dict())
    for m in (g, ref):
        for b in m.modules():
            if isinstance(b, torch.nn.BatchNorm2d):
                b.reset_running_stats()
    monkeypatch.setattr(block, "ENABLED", False)
    want = ref(x)
    monkeypatch.setattr(block, "ENABLED", True)
    got = g(x)
    assert torch.equal(got, want)
    # (2) storage moved under the same Parameter object
    g.fc2[1].bias.data = g.fc2[1].bias.data.clone()
    assert block.try_grapher(g, x) is None
    g(x)
    assert block.try_grapher(g, x) is not None
    # (3) a BN switched to eval mode: outside the driver's scope altogether
    g.fc2[1].eval()
    assert block.try_grapher(g, x) is None
    g.fc2[1].train()
    # (4) DropPath switched on
    from gkgnet_amd.layers import DropPath
    g.drop_path = DropPath(0.5)
    assert block.try_grapher(g, x) is None


def _dp_worker(rank, world, store_path, result_path):
    """Two data-parallel ranks (sharing GPU 0 over gloo): the block driver under the chunked gradient bucket whose all-reduces start
    from hooks DURING the backward (the driver's weight gradients sit in the pass-wide batched launch until a chunk is about to be
    read), against the composition in the same mode and against driver + pack() + one flat all-reduce."""
    import torch.distributed as dist
    from gkgnet_amd import block, parallel
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", store=dist.FileStore(store_path, world), rank=rank, world_size=world)
    try:
        C, H, L, B, G = 64, 12, 20, 16, 2
        gen = torch.Generator().manual_seed(5)
        data = [t.cuda() for t in (torch.randn(world * B, C, H, H, generator=gen), torch.randn(world * B, L, C, generator=gen),
                                   torch.randn(world * B, C, H, H, generator=gen), torch.randn(world * B, L, C, generator=gen))]
        sl = slice(rank * B, (rank + 1) * B)
        results = {}
        for mode in ("driver+hooks", "composition+hooks", "driver+flat"):
            block.ENABLED = not mode.startswith("composition")
            torch.manual_seed(11)
            g = Grapher(C, 9, 2, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                        num_group=G).cuda().train()
            gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                              use_multi_group=True, num_group=G).cuda().train()
            params = list(g.parameters()) + list(gl.parameters())
            bucket = parallel.GradBucket(params, bucket_bytes=16 << 10)             # several chunks
            assert len(bucket.chunks) > 3
            if mode.endswith("hooks"):
                bucket.install_overlap_hooks()
            for step in range(3):                                                    # (dual layout / prepared keys from step 1)
                x, e = data[0][sl].clone().requires_grad_(True), data[1][sl].clone().requires_grad_(True)
                bucket.release(prezero=True)
                out = g(x)
                e2, _ = gl(e, out)
                torch.autograd.backward([out, e2], [data[2][sl], data[3][sl]])
                if mode.endswith("hooks"):
                    bucket.wait()
                else:
                    bucket.pack()
                    bucket.all_reduce()
                torch.cuda.synchronize()
            results[mode] = (bucket.flat.clone(), x.grad.clone(), e.grad.clone())
        ref = results["composition+hooks"]
        for mode in ("driver+hooks", "driver+flat"):
            got = results[mode]
            scale = float(ref[0].abs().max())
            assert float((got[0] - ref[0]).abs().max()) <= 2e-4 * scale, (mode, float((got[0] - ref[0]).abs().max()), scale)
            assert torch.allclose(got[1], ref[1], atol=1e-5, rtol=1e-4) and torch.allclose(got[2], ref[2], atol=1e-5, rtol=1e-4), mode
        other = results["driver+hooks"][0].clone()
        dist.broadcast(other, 0)                                                     # every rank holds the same averaged gradients
        assert torch.equal(other, results["driver+hooks"][0])
        with open(result_path + f".{rank}", "w") as fh:
            fh.write("ok")
    finally:
        block.ENABLED = True
        dist.destroy_process_group()


def test_block_driver_under_the_overlapped_gradient_bucket_two_ranks():
    import os
    import tempfile
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        store, res = os.path.join(d, "store"), os.path.join(d, "res")
        mp.spawn(_dp_worker, args=(2, store, res), nprocs=2, join=True)
        assert all(os.path.exists(res + f".{r}") for r in range(2))


def test_fuzz_block_driver_random_shapes(monkeypatch):
    """Random eligible block pairs (channels, groups, token grid, label count, batch, list size, dilation, with / without a
    positional bias): two steps through the driver and through the composition (~25 s).  The shapes cover the
    one-kernel and the two-launch graph forms, the split-K projection forms and ragged tiles."""
    import time
    import numpy as np
    rng = np.random.RandomState(7)
    t0, done, kinds = time.time(), 0, set()
    from gkgnet_amd import block, fused
    from gkgnet_amd.grapher import Grapher, GrapherLabel

    def run(driver, cfg):
        C, G, H, L, B, k, d, rp = cfg
        monkeypatch.setattr(block, "ENABLED", driver)
        torch.manual_seed(5)
        g = Grapher(C, k, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=rp, use_multi_group=True,
                    num_group=G).cuda().train()
        gl = GrapherLabel(C, k, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                          use_multi_group=True, num_group=G).cuda().train()
        gen = torch.Generator(device="cuda").manual_seed(9)
        res = []
        for step in range(2):
            x = torch.randn(B, C, H, H, device="cuda", generator=gen).requires_grad_(True)
            e = torch.randn(B, L, C, device="cuda", generator=gen).requires_grad_(True)
            for p in list(g.parameters()) + list(gl.parameters()):
                p.grad = None
            out = g(x)
            e2, edge = gl(e, out)
            cx, ce = torch.randn(B, C, H, H, device="cuda", generator=gen), torch.randn(B, L, C, device="cuda", generator=gen)
            torch.autograd.backward([out, e2], [cx, ce])          # (random cotangents: with ones the BN parameter gradients are
            res.append((out.detach(), e2.detach(), edge, x.grad, e.grad,   # mathematically zero, i.e. pure rounding noise)
                        [p.grad.clone() for p in list(g.parameters()) + list(gl.parameters()) if p.grad is not None and p.dim() == 1]))
        torch.cuda.synchronize()
        return res

    def same(a, b):
        """Equal up to what a rounding tie of one BN mean can do (one ulp in one channel, then a near-tie neighbour at worst):
        bit-identity itself is the subject of the test above, which repeats a pair that hit such a tie."""
        for step, (sa, sb) in enumerate(zip(a, b)):
            for i, (u, v) in enumerate(zip(sa[:5], sb[:5])):
                if i == 2:                                                 # the returned graph
                    if float((u == v).float().mean()) < 0.995:
                        return (step, "edge", float((u == v).float().mean()))
                elif float((u - v).abs().max()) > 1e-4 * float(v.abs().max()) + 1e-6:
                    return (step, ("out", "e2", "edge", "dx", "de")[i], float((u - v).abs().max()), float(v.abs().max()))
            # BN parameter gradients against the largest of them: some are mathematically zero (the shift of a BN whose output only
            # feeds another train-mode BN), i.e. rounding noise that a flipped tie re-rolls
            scale = max(float(v.abs().max()) for v in sb[5])
            for j, (u, v) in enumerate(zip(sa[5], sb[5])):
                if float((u - v).abs().max()) > 1e-3 * scale + 1e-5:
                    return (step, "1-D gradient", j, float((u - v).abs().max()), float(v.abs().max()), scale)
        return None

    while time.time() - t0 < 25.0:
        G = int(rng.choice([1, 2, 4]))
        C = 16 * int(rng.randint(1, 13))
        if C % G or (C // G) % 4:
            continue
        H = int(rng.randint(5, 21))
        k = int(rng.choice([3, 5, 9, 12]))
        d = int(rng.randint(1, 4))
        L = int(rng.randint(max(4, 1), 60))
        if k * d > min(H * H, 36) or k > H * H:
            continue
        B = int(rng.randint(1, 9))
        cfg = (C, G, H, L, B, k, d, bool(rng.rand() < 0.6))
        calls = []
        real = block._GrapherBlockFn.forward
        monkeypatch.setattr(block._GrapherBlockFn, "forward", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
        ok, why = False, None
        for attempt in range(2):                                          # (a near-tie neighbour flipped by such a tie)
            on = run(True, cfg)
            if not calls:
                break                                                     # outside the driver's scope (e.g. a projection rule)
            why = same(on, run(False, cfg))
            if why is None:
                ok = True
                break
        monkeypatch.setattr(block._GrapherBlockFn, "forward", staticmethod(real))
        if not calls:
            continue
        assert ok, (cfg, why)
        done += 1
        kinds.add((bool(fused._knn_mr_shapes_ok(B, H * H, C, H * H, False, None, k, d, G, [0, 0, 0], False))))
    assert done >= 8, done
