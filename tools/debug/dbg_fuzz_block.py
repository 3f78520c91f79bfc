"""Driver vs composition on one block-pair configuration, element-level report (debug aid for tests/test_hip_block_driver.py)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from gkgnet_amd import block, fused
from gkgnet_amd.grapher import Grapher, GrapherLabel

def run(driver, cfg, steps=2):
    C, G, H, L, B, k, d, rp = cfg
    block.ENABLED = driver
    torch.manual_seed(5)
    g = Grapher(C, k, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=rp, use_multi_group=True, num_group=G).cuda().train()
    gl = GrapherLabel(C, k, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L, use_multi_group=True, num_group=G).cuda().train()
    gen = torch.Generator(device="cuda").manual_seed(9)
    res = []
    names = [n for n, _ in list(g.named_parameters()) + list(gl.named_parameters())]
    for step in range(steps):
        x = torch.randn(B, C, H, H, device="cuda", generator=gen).requires_grad_(True)
        e = torch.randn(B, L, C, device="cuda", generator=gen).requires_grad_(True)
        for p in list(g.parameters()) + list(gl.parameters()):
            p.grad = None
        out = g(x)
        e2, edge = gl(e, out)
        torch.autograd.backward([out, e2], [torch.ones_like(out), torch.ones_like(e2)])
        torch.cuda.synchronize()
        res.append(dict(out=out.detach(), e2=e2.detach(), edge=edge, dx=x.grad, de=e.grad,
                        **{"grad." + n: (None if p.grad is None else p.grad.clone()) for n, p in zip(names, list(g.parameters()) + list(gl.parameters()))}))
    return res

cfg = eval(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "seq" else (160, 4, 16, 35, 3, 5, 1, True)
for trial in range(0 if (len(sys.argv) > 1 and sys.argv[1] == "seq") else 3):
    a, b = run(True, cfg), run(False, cfg)
    for s, (u, v) in enumerate(zip(a, b)):
        for key in u:
            if u[key] is None or v[key] is None:
                if (u[key] is None) != (v[key] is None): print("trial", trial, "step", s, key, "presence differs")
                continue
            if not torch.equal(u[key], v[key]):
                dlt = (u[key].float() - v[key].float()).abs()
                print("trial", trial, "step", s, key, tuple(u[key].shape), "differs:", int((u[key] != v[key]).sum()), "max", float(dlt.max()), "ref max", float(v[key].float().abs().max()))
    print("trial", trial, "done", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "seq":
    import numpy as np
    rng = np.random.RandomState(7)
    n = 0
    while n < 130:
        G = int(rng.choice([1, 2, 4])); C = 16 * int(rng.randint(1, 13))
        if C % G or (C // G) % 4: continue
        H = int(rng.randint(5, 21)); k = int(rng.choice([3, 5, 9, 12])); d = int(rng.randint(1, 4)); L = int(rng.randint(4, 60))
        if k * d > min(H * H, 36) or k > H * H: continue
        B = int(rng.randint(1, 9)); cfg = (C, G, H, L, B, k, d, bool(rng.rand() < 0.6)); n += 1
        a, b = run(True, cfg), run(False, cfg)
        for s, (u, v) in enumerate(zip(a, b)):
            for key in u:
                if u[key] is None or v[key] is None:
                    continue
                if not torch.equal(u[key], v[key]):
                    dlt = (u[key].float() - v[key].float()).abs()
                    rel = float(dlt.max()) / (float(v[key].float().abs().max()) + 1e-30)
                    if rel > 1e-5 or key == "edge":
                        print(n, cfg, "step", s, key, tuple(u[key].shape), "differs:", int((u[key] != v[key]).sum()), "max", float(dlt.max()), "ref max", float(v[key].float().abs().max()), flush=True)
    print("seq done", n)
