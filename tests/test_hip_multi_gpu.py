"""Two-process RCCL tests (one process per GPU, backend 'nccl' = RCCL over xGMI).  They need >= 2 devices and are SKIPPED
on a single-GPU box — the 1-GPU CI lease the round's tests run on; the same code paths are exercised there over gloo
(tests/test_parallel_cpu.py, tests/test_hip_syncbn.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses duplicate devices)")


@needs_two
@pytest.mark.parametrize("mode", ["flat", "overlap"])
def test_gradient_bucket_over_rccl(mode):
    from test_parallel_cpu import _run
    _run(mode, backend="nccl")


@needs_two
def test_fused_syncbn_over_rccl():
    """The fused block's cross-rank batch statistics (all-reduced column sums) over RCCL: two ranks with half a batch
    each against the single-process full batch."""
    import test_hip_syncbn as T
    T.run_two_ranks(backend="nccl")
