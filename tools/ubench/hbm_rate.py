#!/usr/bin/env python
"""What a plain streaming pass achieves on this GPU (torch copy / fill / add, 400 MB operands): the practical ceiling the
gather / scatter kernels' 'fraction of 8 TB/s' should be read against."""
import torch
n = 100 * 1024 * 1024
a = torch.randn(n, device="cuda"); b = torch.empty_like(a); c = torch.randn(n, device="cuda")


def t(fn, bytes_):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    return f"{us:7.1f} us  {bytes_ / us / 1e6:5.2f} TB/s"


print("copy  (read 400 MB + write 400 MB):", t(lambda: b.copy_(a), 8 * n))
print("fill  (write 400 MB)              :", t(lambda: b.fill_(1.0), 4 * n))
print("add   (read 800 MB + write 400 MB):", t(lambda: torch.add(a, c, out=b), 12 * n))
print("sum   (read 400 MB)               :", t(lambda: a.sum(), 4 * n))
