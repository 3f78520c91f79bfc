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


@pytest.mark.parametrize("sync_bn", [False, True])
def test_bench_starts_its_own_ranks_without_a_launcher(sync_bn):
    """``python bench.py --gpus 2 [--sync-bn]`` with no WORLD_SIZE in the environment: the parent spawns the two ranks itself
    (it never touches the GPU), relays rank 0's JSON line and exits with the children's code.  On a 1-GPU box the two ranks
    share the device over gloo (GKG_DIST_BACKEND=gloo; RCCL refuses duplicate devices) — the N > 1 control flow of the
    driver's scaling run: shard seeds, flat gradient all-reduce, max-over-ranks timing; with --sync-bn the cross-rank batch
    statistics of every BN layer inside the fused blocks (the reference's DDP semantics).  The JSON line names the world
    size, the backend and every rank's device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two = torch.cuda.device_count() >= 2
    if not two:
        env["GKG_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-tune",
           "--no-cpu-baseline", "--batch", "4"] + (["--sync-bn"] if sync_bn else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 2 and res["steps"] == 3 and cfg["global_batch"] == 8
    assert res["value"] > 0 and res["scaling"] == "weak" and cfg["parallelism"] == "dp2"
    assert cfg["world_size"] == 2 and cfg["backend"] == ("nccl" if two else "gloo")
    assert len(cfg["devices"]) == 2 and cfg["devices"][0].startswith("rank 0: cuda:") and cfg["devices"][1].startswith("rank 1: cuda:")
    assert cfg["bn"] == ("sync" if sync_bn else "local")
    if two:
        assert "captured inside the step's hipGraph" in cfg["grad_allreduce"]
    else:
        assert "issued by the host" in cfg["grad_allreduce"]
