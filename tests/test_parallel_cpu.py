"""world_size-2 gloo test of the data-parallel plumbing (flat gradient bucket + single all-reduce)."""
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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gkgnet_amd.parallel import GradBucket, broadcast_parameters, init_distributed, shard_batch
    init_distributed("gloo")
    torch.manual_seed(100 + rank)                       # ranks start different ...
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 1), torch.nn.BatchNorm2d(6), torch.nn.Conv2d(6, 2, 1))
    broadcast_parameters(net)                           # ... and are made identical
    bucket = GradBucket(net.parameters())
    g = torch.Generator().manual_seed(0)
    data = torch.randn(6, 4, 3, 3, generator=g)         # global batch, same on both ranks
    mine = data[list(shard_batch(6, rank, world))]
    bucket.release()
    net(mine).square().sum().backward()
    bucket.pack()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    bucket.all_reduce()
    out[rank] = (bucket.flat.clone(), [p.detach().clone() for p in net.parameters()])
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_manual_average():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    f0, p0 = out[0]
    f1, p1 = out[1]
    assert torch.equal(f0, f1)                          # identical averaged gradients everywhere
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    # manual reference: average of the two per-shard gradients
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 1), torch.nn.BatchNorm2d(6), torch.nn.Conv2d(6, 2, 1))
    with torch.no_grad():
        for p, v in zip(net.parameters(), p0):
            p.copy_(v)
    g = torch.Generator().manual_seed(0)
    data = torch.randn(6, 4, 3, 3, generator=g)
    grads = []
    for sl in (slice(0, 3), slice(3, 6)):
        net.zero_grad()
        net(data[sl]).square().sum().backward()
        grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]))
    assert torch.allclose(f0, (grads[0] + grads[1]) / 2, atol=1e-6)


def test_shard_batch_partitions():
    from gkgnet_amd.parallel import shard_batch
    for gb, w in ((256, 8), (10, 4), (3, 8)):
        seen = [i for r in range(w) for i in shard_batch(gb, r, w)]
        assert seen == list(range(gb))
