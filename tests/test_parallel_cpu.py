"""world_size-2 gloo tests of the data-parallel plumbing: flat gradient bucket with one all-reduce, and the chunked form
whose all-reduces start from post-accumulate hooks during the backward (the reference's DDP reducer, apis/train.py:117-125)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    return torch.nn.Sequential(torch.nn.Conv2d(4, 6, 1), torch.nn.BatchNorm2d(6), torch.nn.Conv2d(6, 2, 1),
                               torch.nn.BatchNorm2d(2), torch.nn.Conv2d(2, 3, 1))


def _worker(rank, world, port, out, mode, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gkgnet_amd.parallel import GradBucket, broadcast_parameters, init_distributed, shard_batch
    init_distributed(backend)
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    torch.manual_seed(100 + rank)                       # ranks start different ...
    net = _net().to(dev)
    broadcast_parameters(net)                           # ... and are made identical
    unused = torch.nn.Parameter(torch.ones(5, device=dev))          # a parameter the loss never touches
    bucket = GradBucket(list(net.parameters()) + [unused], bucket_bytes=64)     # tiny chunks: several collectives
    assert len(bucket.chunks) > 2
    g = torch.Generator().manual_seed(0)
    data = torch.randn(6, 4, 3, 3, generator=g).to(dev)  # global batch, same on both ranks
    mine = data[list(shard_batch(6, rank, world))]
    for step in range(2):                               # two steps: hooks / views must survive re-use
        bucket.release()
        if mode == "overlap":
            bucket.install_overlap_hooks()
            net(mine).square().sum().backward()
            bucket.wait()
        else:
            net(mine).square().sum().backward()
            bucket.pack()
            bucket.all_reduce()
        lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi for p in bucket.params)
        assert float(unused.grad.abs().sum()) == 0.0
    out[rank] = ([p.grad.detach().cpu().clone() for p in net.parameters()], [p.detach().cpu().clone() for p in net.parameters()])
    dist.destroy_process_group()


def _run(mode, backend="gloo"):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, mode, backend), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                        # identical averaged gradients everywhere
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    # manual reference: average of the two per-shard gradients
    net = _net()
    with torch.no_grad():
        for p, v in zip(net.parameters(), p0):
            p.copy_(v)
    g = torch.Generator().manual_seed(0)
    data = torch.randn(6, 4, 3, 3, generator=g)
    grads = []
    for sl in (slice(0, 3), slice(3, 6)):
        net.zero_grad()
        net(data[sl]).square().sum().backward()
        grads.append([p.grad.clone() for p in net.parameters()])
    for got, a, b in zip(g0, grads[0], grads[1]):
        assert torch.allclose(got, (a + b) / 2, atol=1e-6)


def test_flat_bucket_allreduce_matches_manual_average():
    _run("flat")


def test_overlapped_chunk_allreduce_matches_manual_average():
    _run("overlap")


def test_grad_view_is_adopted_without_copy():
    """A backward that returns ``grad_view(p)`` (what the fused blocks do for weight / BN gradients) leaves ``p.grad``
    aliasing the bucket: pack() has nothing to move for it."""
    from gkgnet_amd.parallel import GradBucket, grad_view

    class WriteInPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            out = grad_view(w)
            assert out is not None
            torch.mm(g.t(), x, out=out)
            return g @ w, out
    w = torch.nn.Parameter(torch.randn(3, 4))
    other = torch.nn.Parameter(torch.randn(2))
    bucket = GradBucket([w, other])
    x = torch.randn(5, 4)
    bucket.release()
    (WriteInPlace.apply(x, w).sum() + other.sum()).backward()
    assert bucket._resident(w) and not bucket._resident(other)
    want = torch.ones(5, 3).t() @ x
    bucket.pack()
    assert bucket._resident(other) and torch.allclose(w.grad, want) and torch.equal(other.grad, torch.ones(2))
    assert grad_view(w) is None                         # grads attached: an accumulating step must not write in place


def test_shard_batch_partitions():
    from gkgnet_amd.parallel import shard_batch
    for gb, w in ((256, 8), (10, 4), (3, 8)):
        seen = [i for r in range(w) for i in shard_batch(gb, r, w)]
        assert seen == list(range(gb))


def test_bucket_slots_of_missing_gradients_stay_zero():
    """Parameters that receive no gradient keep a zero slot without a fill per step; a slot that held a gradient is cleared
    again when the gradient disappears."""
    import torch
    from gkgnet_amd.parallel import GradBucket
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(5))
    bucket = GradBucket([a, b])
    for step, use_b in enumerate([False, True, False, False]):
        bucket.release()
        loss = (a * 2).sum() + ((b * 3).sum() if use_b else 0)
        loss.backward()
        bucket.pack()
        assert torch.equal(a.grad, torch.full((3,), 2.0)), step
        assert torch.equal(b.grad, torch.full((5,), 3.0 if use_b else 0.0)), step
        assert a.grad.data_ptr() == bucket._view(a).data_ptr() and b.grad.data_ptr() == bucket._view(b).data_ptr()


def test_wait_after_a_step_without_backward_clears_the_gradients():
    """a chunk whose hooks never fired must not be mistaken for a finished one — wait() fills it (zeros) instead
    of re-attaching the previous step's reduced gradients."""
    from gkgnet_amd.parallel import GradBucket
    w = torch.nn.Parameter(torch.randn(4, 4))
    v = torch.nn.Parameter(torch.randn(3))
    bucket = GradBucket([w, v], bucket_bytes=16)
    bucket.install_overlap_hooks()
    bucket.release()
    ((w * w).sum() + v.sum()).backward()
    bucket.wait()
    assert float(w.grad.abs().sum()) > 0
    bucket.release()                      # a step with no backward at all
    bucket.wait()
    assert float(w.grad.abs().sum()) == 0.0 and float(v.grad.abs().sum()) == 0.0
    bucket.release()                      # only one of the two parameters used
    v.sum().backward()
    bucket.wait()
    assert float(w.grad.abs().sum()) == 0.0 and torch.equal(v.grad, torch.ones(3))


def _divergent_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gkgnet_amd.parallel import GradBucket, init_distributed
    init_distributed("gloo")
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 9)]
    bucket = GradBucket(ps, bucket_bytes=8, find_unused_parameters=True)     # one chunk per parameter, of different sizes
    assert len(bucket.chunks) == 4
    bucket.install_overlap_hooks()
    for step in range(2):
        bucket.release()
        # rank 0 uses parameters 0 and 2, rank 1 uses 1 and 3: the chunks complete in different orders on the two ranks
        # and each rank finishes half of them only inside wait()
        use = (0, 2) if rank == 0 else (1, 3)
        sum((ps[i] * (i + 1)).sum() for i in use).backward()
        bucket.wait()
    out[rank] = [p.grad.clone() for p in ps]
    dist.destroy_process_group()


def test_chunks_are_reduced_in_fixed_order_when_ranks_use_different_parameters():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_divergent_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for i, (a, b) in enumerate(zip(out[0], out[1])):
        assert torch.equal(a, b)
        assert torch.allclose(a, torch.full_like(a, (i + 1) / 2.0))     # one rank contributed i+1, the other 0


def test_grad_view_is_handed_out_once_per_backward():
    """a weight used twice in one graph — the second backward node must not overwrite the slot the first one
    wrote; it gets a fresh tensor that autograd adds."""
    from gkgnet_amd.parallel import GradBucket, grad_view

    class WriteInPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            out = grad_view(w)
            if out is None:
                out = torch.empty_like(w)
            torch.mm(g.t(), x, out=out)
            return g @ w, out
    w = torch.nn.Parameter(torch.randn(4, 4))
    bucket = GradBucket([w])
    x1, x2 = torch.randn(5, 4), torch.randn(6, 4)
    for _ in range(2):
        bucket.release()
        (WriteInPlace.apply(x1, w).sum() + WriteInPlace.apply(x2, w).sum()).backward()
        bucket.pack()
        want = torch.ones(5, 4).t() @ x1 + torch.ones(6, 4).t() @ x2
        assert torch.allclose(w.grad, want, atol=1e-5)
        assert bucket._resident(w)


def _no_release_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gkgnet_amd.parallel import GradBucket, init_distributed
    init_distributed("gloo")
    ps = [torch.nn.Parameter(torch.ones(n)) for n in (4, 6, 3)]
    bucket = GradBucket(ps, bucket_bytes=8)
    bucket.install_overlap_hooks()
    seen = []
    for step in range(3):
        bucket.zero()                                   # gradients stay attached: the backward accumulates into the bucket
        sum((p * float(rank + 1)).sum() for p in ps).backward()
        bucket.wait()
        seen.append([p.grad.clone() for p in ps])
    out[rank] = seen
    dist.destroy_process_group()


def test_overlap_hooks_reduce_every_step_without_release():
    """wait() must re-arm the chunk cursor — a zero() + backward + wait() loop that never calls release() has to
    average the gradients in EVERY step (rank contributions 1 and 2 -> 1.5), not only in the first."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_no_release_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for step in range(3):
        for a, b in zip(out[0][step], out[1][step]):
            assert torch.equal(a, b), step
            assert torch.allclose(a, torch.full_like(a, 1.5)), (step, a)


def test_release_prezero_marks_slots_clean_once():
    """release(prezero=True): ONE fill of the flat buffer; every slot handed out by grad_view() in the following backward
    carries the "holds zeros" mark exactly once (atomically accumulating kernels skip their own zero-fill), slots of
    parameters without a gradient already hold what pack() would write, and a plain release() hands out unmarked views."""
    from gkgnet_amd.parallel import GradBucket, grad_view
    w, v, unused = (torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2)))
    bucket = GradBucket([w, v, unused])
    bucket.flat.fill_(7.0)                              # stale contents
    bucket.release(prezero=True)
    assert float(bucket.flat.abs().sum()) == 0.0
    a = grad_view(w)
    assert getattr(a, "_gkg_zero", False) and float(a.abs().sum()) == 0.0
    assert grad_view(w) is None                         # handed out once per backward
    a.add_(1.0)                                         # an "atomic accumulation" into the clean slot
    w.grad = a
    v.grad = torch.full((5,), 2.0)
    bucket.pack()
    assert torch.equal(w.grad, torch.ones(3, 4)) and torch.equal(v.grad, torch.full((5,), 2.0))
    assert float(unused.grad.abs().sum()) == 0.0
    bucket.release()                                    # no pre-zero: the view is not marked (its contents are stale)
    b = grad_view(w)
    assert b is not None and not getattr(b, "_gkg_zero", False)


def test_bucket_clip_grad_norm_matches_torch():
    """GradBucket.clip_grad_norm_ (norm + scale on the flat buffer) against torch.nn.utils.clip_grad_norm_ on a copy:
    clipping and non-clipping thresholds, with one parameter that receives no gradient."""
    import copy
    from gkgnet_amd.parallel import GradBucket
    for max_norm in (0.05, 1e6):
        torch.manual_seed(3)
        net = _net()
        ref = copy.deepcopy(net)
        x = torch.randn(5, 4, 3, 3)
        bucket = GradBucket(net.parameters())
        bucket.release()
        net[:4](x).square().mean().backward()               # the last conv stays unused: its slots hold zeros
        bucket.pack()
        total = bucket.clip_grad_norm_(max_norm)
        ref[:4](x).square().mean().backward()
        used = [p for p in ref.parameters() if p.grad is not None]
        want = torch.nn.utils.clip_grad_norm_(used, max_norm)
        torch.testing.assert_close(total, want, rtol=1e-6, atol=0)
        for p, q in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p.grad, q.grad if q.grad is not None else torch.zeros_like(q), rtol=1e-6, atol=1e-12)
