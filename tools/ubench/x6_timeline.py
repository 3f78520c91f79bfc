#!/usr/bin/env python
"""Workgroup phase timeline of gemm_x6_kernel (s_memtime stamps of every 16th workgroup's first wave; a -DX6_TIMELINE build
of the library in /tmp): where a workgroup's life goes at the short-K stage-1 / stage-2 projection shapes, which run at
about half the practical HBM rate.    python tools/ubench/x6_timeline.py"""
import ctypes, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gkgnet_amd import _build


def build_tl():
    so = "/tmp/libgkg_x6tl.so"
    os.makedirs("/tmp/x6tl", exist_ok=True)
    objs = [f"/tmp/x6tl/{s}.o" for s in _build.SOURCES]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda so_: subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DX6_TIMELINE", "-c", os.path.join(_build.CSRC, so_[0]), "-o", so_[1]]),
                    zip(_build.SOURCES, objs)))
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    return so


os.environ["GKG_HIP_LIB"] = build_tl()
import numpy as np
import torch
from gkgnet_amd import _lib

lib = _lib.load()
lib.gkg_debug_set_x6_timeline.argtypes = [ctypes.c_void_p]
SHAPES = [(663552, 80, 80), (663552, 160, 80), (663552, 80, 320), (663552, 320, 80), (165888, 160, 160), (165888, 160, 640),
          (41472, 400, 400)]
PHASES = ["issue DMA", "wait 1st stage", "K loop", "stores issued", "stores done"]


def planes(w, cout, cin):
    pf = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, 1, 0), dtype=torch.uint8, device="cuda")
    host = ctypes.create_string_buffer(lib.gkg_x6_prep_desc_bytes())
    units = lib.gkg_x6_prep_desc_fill(host, 0, w.data_ptr(), pf.data_ptr(), None, cin, cout, 1, 0, 0)
    descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).cuda()
    _lib.check(lib.gkg_x6_prep_weights(descs.data_ptr(), 1, units, None), "prep")
    return pf


flush = torch.empty(128 << 20, dtype=torch.float32, device="cuda")
for R, cin, cout in SHAPES:
    x = torch.randn(R, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") * 0.1
    y = torch.empty(R, cout, device="cuda")
    pf = planes(w, cout, cin)
    tl = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    lib.gkg_debug_set_x6_timeline(tl.data_ptr())

    def call():
        _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, 1, 0,
                                            *([None] * 10), 0.0, 0.0, None, None), "fwd")
    for _ in range(3):
        call()
    flush.add_(1.0)
    tl.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = tl.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    d = np.diff(t[:, :6], axis=1)
    life = t[:, 5] - t[:, 0]
    nwg = ((R + 127) // 128 + 7) // 8 * 8 * ((cout + 63) // 64)
    print(f"R={R} {cin}->{cout}: {us:.1f} us (one eager launch); {nwg} workgroups, sampled {len(t)}; workgroup life in cycles: "
          f"median {int(np.median(life))} (p10 {int(np.percentile(life, 10))}, p90 {int(np.percentile(life, 90))})")
    for i, name in enumerate(PHASES):
        print(f"     {name:16s} median {int(np.median(d[:, i])):6d}  p10 {int(np.percentile(d[:, i], 10)):6d}  p90 {int(np.percentile(d[:, i], 90)):6d}")
lib.gkg_debug_set_x6_timeline(None)
