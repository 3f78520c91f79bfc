"""cProfile of the eager cfg2 step (host side): where the Python time between launches goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gkgnet_amd import parallel

w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
grapher, label = bench.build_modules(w, dev)
params = list(grapher.parameters()) + list(label.parameters())
bucket = parallel.GradBucket(params)
B, C, H, L = 32, w["C"], w["H"], w["L"]
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

for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 50 * 1e3)
if len(sys.argv) > 1 and sys.argv[1] == "time":
    sys.exit(0)
# backward on the calling thread, so that the profile sees the custom Functions' backward methods too
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumtime").print_stats("gkgnet_amd", 40)
print(s.getvalue()[:9000])
