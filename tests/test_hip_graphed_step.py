"""gkgnet_amd.graphed.GraphedStep: a Grapher -> GrapherLabel training step (reference torch_vertex.py:325-333 -> :392-403, loop
mmcls/apis/train.py:117-180) captured into a hipGraph gives what the eager step gives, batch after batch, with an in-graph
optimiser step; a step that cannot be captured falls back to eager launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build():
    from gkgnet_amd import parallel
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    torch.manual_seed(3)
    C, H, L, B = 64, 12, 20, 4
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=2).cuda().train()
    params = list(g.parameters()) + list(gl.parameters())
    bucket = parallel.GradBucket(params)
    opt = torch.optim.SGD(params, lr=0.05)
    x = torch.zeros(B, C, H, H, device="cuda")
    e = torch.zeros(B, L, C, device="cuda")
    loss = torch.zeros((), device="cuda")

    def step():
        bucket.release(prezero=True)
        out = g(x)
        e2, _ = gl(e, out)
        val = (out.float() ** 2).mean() + (e2.float() ** 2).mean()
        val.backward()
        bucket.pack()
        opt.step()
        loss.copy_(val.detach())
    return g, gl, x, e, loss, step


def _batches(n):
    gen = torch.Generator(device="cuda").manual_seed(11)
    return [(torch.randn(4, 64, 12, 12, device="cuda", generator=gen), torch.randn(4, 20, 64, device="cuda", generator=gen))
            for _ in range(n)]


def test_replayed_steps_follow_the_eager_ones():
    from gkgnet_amd.graphed import GraphedStep
    data = _batches(6)
    # eager reference: the 3 warm-up steps on the first batch (the capturing call records the step, it does not run it), then one
    # step per batch
    g, gl, x, e, loss, step = _build()
    x.copy_(data[0][0]); e.copy_(data[0][1])
    for _ in range(3):
        step()
    ref = []
    for bx, be in data[1:]:
        x.copy_(bx); e.copy_(be)
        step()
        ref.append(float(loss))
    ref_w = [p.detach().clone() for p in list(g.parameters()) + list(gl.parameters())]
    # the same through a captured step
    g2, gl2, x2, e2, loss2, step2 = _build()
    x2.copy_(data[0][0]); e2.copy_(data[0][1])
    gs = GraphedStep(step2, warmup=3)
    assert gs.captured
    got = []
    for bx, be in data[1:]:
        x2.copy_(bx); e2.copy_(be)
        gs.replay()
        got.append(float(loss2))
    for a, b in zip(got, ref):
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-6, (got, ref)
    for p, q in zip(list(g2.parameters()) + list(gl2.parameters()), ref_w):
        assert float((p.detach() - q).abs().max()) <= 2e-3 * float(q.abs().max()) + 1e-5


def test_a_step_that_cannot_be_captured_runs_eagerly(capsys):
    from gkgnet_amd.graphed import GraphedStep
    t = torch.zeros(4, device="cuda")
    calls = []

    def step():
        t.add_(1.0)
        calls.append(float(t[0]))            # a host read: synchronises, which a capture does not allow

    gs = GraphedStep(step, warmup=1)
    assert not gs.captured
    assert "run eagerly" in capsys.readouterr().err
    before = float(t[0])
    gs.replay()
    assert float(t[0]) == before + 1.0
    with pytest.raises(ValueError):
        GraphedStep(step, warmup=0)
