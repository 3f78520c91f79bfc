"""Replay a training step from a hipGraph: what ``bench.py`` times, as a helper for a caller's own loop.

A ``Grapher`` + ``GrapherLabel`` step at 18 x 18 tokens is ~50 launches of 5-30 us each: launched op by op from Python it is
host-bound (1.7-1.9 ms per step at B = 32 on the measurement host against 0.82-0.90 ms of GPU work).  The step has
static shapes and no host synchronisation, so it can be captured once and replayed — the reference's training loop
(mmcls/apis/train.py:117-180: forward, loss, backward, optimiser step per iteration) keeps its structure, only the body of the
iteration becomes ``copy the batch into the static inputs; step.replay()``.

Rules the captured function must follow (the library checks what it can and raises ``GkgError`` otherwise):
  * it reads its inputs from tensors that stay allocated (copy each batch INTO them) and leaves its results in tensors the caller
    reads after ``replay()``;
  * gradients go through a :class:`gkgnet_amd.parallel.GradBucket` (``bucket.release(prezero=True)`` first, ``bucket.pack()``
    last) or through ``.grad`` tensors that already exist — a capture cannot allocate ``.grad`` on the fly;
  * one eager call comes first (``warmup`` >= 1): weight planes, BN scratch and workspaces are created outside captures.
"""
from __future__ import annotations

import sys
from typing import Callable, Optional

import torch


class GraphedStep:
    """``GraphedStep(fn, warmup=3)`` runs ``fn()`` ``warmup`` times on a side stream (eager), captures one more call into a
    hipGraph and keeps it; ``replay()`` launches the captured step (one graph launch).  If the capture fails — something in ``fn``
    synchronises with the host, allocates under capture, or a process group's collective is not capturable — the error is reported
    once on stderr and ``replay()`` calls ``fn()`` eagerly instead (``captured`` tells which)."""

    def __init__(self, fn: Callable[[], None], warmup: int = 3, fallback: bool = True):
        if warmup < 1:
            raise ValueError("GraphedStep: at least one eager warm-up call is required (weight planes, BN scratch and workspaces are "
                             "created on first use, which must not happen inside a capture)")
        self.fn = fn
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: another thread of the process (a collective's watchdog) may touch the device while this one captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self.graph = g
        except Exception as exc:
            if not fallback:
                raise
            print(f"[gkgnet_amd] hipGraph capture of the step failed ({type(exc).__name__}: {exc}); it will run eagerly",
                  file=sys.stderr)
            torch.cuda.synchronize()
            fn()                            # leave the caller's buffers in a defined state
            torch.cuda.synchronize()

    @property
    def captured(self) -> bool:
        return self.graph is not None

    def replay(self):
        if self.graph is None:
            self.fn()
        else:
            self.graph.replay()

    __call__ = replay
