#!/usr/bin/env python
"""Per-shape timing of the x6 projection kernels (forward store-only / forward + BN statistics / dgrad) for each column
tile width against the vendor fp32 GEMM (torch.mm / bmm, TunableOp off), with the operands evicted from L2 between
launches by a 512 MB sweep (in-step conditions).    python tools/bench_x6.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib, fused

lib = _lib.load()
SHAPES = [(10368, 320, 320, 1), (10368, 160, 160, 4), (10368, 640, 320, 1), (10368, 320, 640, 1), (2560, 320, 320, 1),
          (2560, 160, 160, 4), (2560, 640, 320, 1), (2560, 320, 1280, 1), (2560, 1280, 320, 1), (41472, 400, 400, 1),
          (41472, 800, 400, 1)]
if len(sys.argv) > 1 and sys.argv[1] in ("cfg4", "wgradcfg4"):        # GKGNet-576 (pvig_s) stage shapes at B = 32
    SHAPES = [(663552, 80, 80, 1), (663552, 160, 80, 1), (663552, 80, 320, 1), (663552, 320, 80, 1), (165888, 160, 160, 1),
              (165888, 320, 160, 1), (165888, 160, 640, 1), (165888, 640, 160, 1), (41472, 400, 400, 1), (41472, 800, 400, 1),
              (41472, 400, 1600, 1), (41472, 1600, 400, 1), (10368, 640, 640, 1), (10368, 1280, 640, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "coltiles":    # one vs two column tiles at the stage-1 row count
    SHAPES = [(663552, 80, 64, 1), (663552, 80, 80, 1), (663552, 80, 128, 1), (663552, 160, 64, 1), (663552, 160, 80, 1),
              (663552, 160, 128, 1), (663552, 320, 64, 1), (663552, 320, 80, 1), (663552, 64, 320, 1), (663552, 80, 320, 1)]
WGRAD_ONLY = len(sys.argv) > 1 and sys.argv[1] in ("wgrad", "wgradcfg4")      # wgradcfg4: the weight gradients at GKGNet-576's stage shapes
X6_ONLY = bool(os.environ.get("X6_ONLY"))               # forward / dgrad of the own kernels only (A/B of two library builds)
if WGRAD_ONLY and sys.argv[1] == "wgrad":               # every weight gradient of the cfg2 step (Grapher rows 10 368, label rows 2 560)
    SHAPES = [(10368, 320, 320, 1), (10368, 160, 160, 4), (10368, 640, 320, 1), (10368, 320, 1280, 1), (10368, 1280, 320, 1),
              (2560, 320, 320, 1), (2560, 160, 160, 4), (2560, 640, 320, 1), (2560, 320, 1280, 1), (2560, 1280, 320, 1)]
flush = torch.empty(128 << 20, dtype=torch.float32, device="cuda")


def st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=20):
    """Mean device time of fn() replayed from a hipGraph (no launch gaps), operands evicted before every replay."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    tot = 0.0
    for _ in range(n):
        flush.add_(1.0)                      # sweep 512 MB through the caches
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        tot += a.elapsed_time(b)
    return 1e3 * tot / n


def planes(w, nb, cout, cin):
    pf = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 0), dtype=torch.uint8, device="cuda")
    pd = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 1), dtype=torch.uint8, device="cuda")
    host = ctypes.create_string_buffer(lib.gkg_x6_prep_desc_bytes())
    units = lib.gkg_x6_prep_desc_fill(host, 0, w.data_ptr(), pf.data_ptr(), pd.data_ptr(), cin, cout, nb, 0, 0)
    descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).cuda()
    _lib.check(lib.gkg_x6_prep_weights(descs.data_ptr(), 1, units, None), "prep")
    return pf, pd


for R, cin, cout, nb in SHAPES:
    x = torch.randn(nb, R, cin, device="cuda"); w = torch.randn(nb, cout, cin, device="cuda") * 0.1
    dy = torch.randn(nb, R, cout, device="cuda")
    y = torch.empty(nb, R, cout, device="cuda"); dx = torch.empty(nb, R, cin, device="cuda")
    pf, pd = planes(w, nb, cout, cin)
    stats = fused._stats_scratch(x.device)
    dw = torch.zeros(nb, cout, cin, device="cuda")
    vw = 0.0 if X6_ONLY else timeit(lambda: fused._wgrad(dy[0], x[0]) if nb == 1 else fused._wgrad_grouped(dy, x))
    xw = timeit(lambda: (dw.zero_(), lib.gkg_linear_wgrad_x6(dy.data_ptr(), cout, R * cout, x.data_ptr(), cin, R * cin, dw.data_ptr(),
                                                              R, cin, cout, nb, 0, st())))
    print(f"R={R:6d} {cin:4d}->{cout:4d} nb={nb}: wgrad vendor (split-K bmm) {vw:6.1f}  x6 (incl. memset) {xw:6.1f}", flush=True)
    if WGRAD_ONLY:
        continue
    line = f"R={R:6d} {cin:4d}->{cout:4d} nb={nb}:"
    if not X6_ONLY:
        line += f" vendor fwd {timeit(lambda: torch.bmm(x, w.transpose(1, 2), out=y)):6.1f} dgrad {timeit(lambda: torch.bmm(dy, w, out=dx)):6.1f} |"
    for ni in ("auto",):
        f = timeit(lambda: lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 0,
                                                    *([None] * 10), 0.0, 0.0, None, st()))
        fs = timeit(lambda: lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 2,
                                                     *([None] * 10), 0.0, 0.0, stats.data_ptr(), st()))
        stats.zero_()
        d = timeit(lambda: lib.gkg_linear_dgrad_x6(dy.data_ptr(), cout, R * cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb, st()))
        line += f" NI={ni}: fwd {f:5.1f} +stats {fs:5.1f} dgrad {d:5.1f} |"
    print(line, flush=True)
