#!/usr/bin/env python
"""Every workgroup of wgrad_x6_batch_kernel in one cfg2 backward: start / end stamp, CU, problem (a -DX6_TIMELINE build of
gkg_gemm_x6.hip linked with the library's other objects): how evenly the launch fills the chip.
python tools/ubench/wgrad_batch_timeline.py build  (here)      python tools/ubench/wgrad_batch_timeline.py  (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "ubench", "_knn_ablate")
SO = os.path.join(OUT, "libgkg_hip_x6tl.so")
if sys.argv[1:] == ["build"]:
    from gkgnet_amd import _build
    _build.build()
    os.makedirs(OUT, exist_ok=True)
    obj = "/tmp/gkg_gemm_x6_tl.o"
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DX6_TIMELINE", "-c", os.path.join(_build.CSRC, "gkg_gemm_x6.hip"), "-o", obj])
    objs = [os.path.join(_build.PKG, "build", s.replace(".hip", ".o")) for s in _build.SOURCES if s != "gkg_gemm_x6.hip"]
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + [obj])
    sys.exit(0)
os.environ["GKG_HIP_LIB"] = SO
import numpy as np
import torch
import bench
from gkgnet_amd import _lib, parallel

lib = _lib.load()
lib.gkg_debug_set_x6_timeline.argtypes = [ctypes.c_void_p]
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
grapher, label = bench.build_modules(w, dev)
params = list(grapher.parameters()) + list(label.parameters())
bucket = parallel.GradBucket(params)
B, C, H, L = w.get("B", 32), w["C"], w["H"], w["L"]
x = torch.randn(B, C, H, H, device=dev).requires_grad_(True)
e = torch.randn(B, L, C, device=dev).requires_grad_(True)
cx, ce = torch.randn(B, C, H, H, device=dev), torch.randn(B, L, C, device=dev)


def step():
    bucket.release(prezero=True)
    x.grad = None; e.grad = None
    out = grapher(x)
    e2, _ = label(e, out)
    torch.autograd.backward([out, e2], [cx, ce])
    bucket.pack()


tl = torch.zeros(4096 * 8 + 4096 * 4 + 4096, dtype=torch.int64, device="cuda")
for _ in range(5):
    step()
torch.cuda.synchronize()
lib.gkg_debug_set_x6_timeline(tl.data_ptr())
tl.zero_()
step()
torch.cuda.synchronize()
full = tl.cpu().numpy()
g = full[4096 * 8:4096 * 12].reshape(4096, 4)
loopdone = full[4096 * 12:]
keep = g[:, 0] > 0
loopdone = loopdone[keep]
g = g[keep]
hw, xcc = g[:, 2] >> 8, g[:, 2] & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
# the cycle counter is only comparable between workgroups of ONE CU: per CU, order its workgroups by start
life = (g[:, 1] - g[:, 0]).astype(np.int64)
prob = g[:, 3] >> 1
real = life > 20000                                  # (the padded grid's workgroups exit at once)
print(f"{len(g)} workgroups ({int(real.sum())} with work) on {len(np.unique(key))} CUs; counter ticks ~ shader clock")
per_cu = {}
for k_, a, b_, ok in zip(key, g[:, 0], g[:, 1], real):
    if ok:
        per_cu.setdefault(int(k_), []).append((int(a), int(b_)))
n_per_cu = np.array([len(v) for v in per_cu.values()])
print("working workgroups per CU:", dict(zip(*[x.tolist() for x in np.unique(n_per_cu, return_counts=True)])), f"; CUs without one: {256 - len(per_cu)}")
spans, gaps, overl = [], [], 0
for iv in per_cu.values():
    iv.sort()
    spans.append(iv[-1][1] - iv[0][0])
    for (a0, b0), (a1, b1) in zip(iv, iv[1:]):
        if a1 < b0:
            overl += 1
        else:
            gaps.append(a1 - b0)
spans = np.array(spans)
print(f"per CU: first start -> last end: median {int(np.median(spans))}, max {spans.max()} ticks; workgroup life: median {int(np.median(life[real]))}, "
      f"min {life[real].min()}, max {life[real].max()}; back-to-back gap on a CU: median {int(np.median(gaps)) if gaps else 0}; overlapping pairs: {overl}")
print(f"chip utilisation if every CU were busy for the longest CU's span: {life[real].sum() / (256.0 * spans.max()):.2f}")
for p_ in np.unique(prob):
    m = (prob == p_) & real
    if m.any():
        print(f"   problem {int(p_):2d} ({'64x128' if (g[m, 3][0] & 1) else '64x64 '} body): {int(m.sum()):3d} workgroups, life min/median/max {life[m].min()}/{int(np.median(life[m]))}/{life[m].max()} ticks")
ep = (g[:, 1] - loopdone)[real & (loopdone > 0)]
lf = life[real & (loopdone > 0)]
if len(ep):
    print(f"epilogue (all waves' K loops done -> end: LDS reduction of the four partial tiles + 8 192 atomics): median {int(np.median(ep))} ticks = "
          f"{np.median(ep / lf):.2f} of a workgroup's life (min {ep.min()}, max {ep.max()})")
