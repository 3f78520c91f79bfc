#!/usr/bin/env python
"""Compile-time phase ablation of the direct first-stem-convolution kernel (csrc/gkg_stem.hip): private copies with
-DSTEM_ABL=<bits> in /tmp, timed with HIP events at the cfg3 shape (B = 32, 3 -> 40, 576 x 576).
    python tools/ubench/stem_ablate.py [variant ...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
VARIANTS = {"full": 0, "no_loads": 1, "no_stores": 2, "no_fma": 4, "loads_only": 6, "stores_only": 5, "fma_only": 3, "skeleton": 7}


def build(bits, extra=()):
    so = f"/tmp/libstem_{bits}_{'_'.join(e.replace('=', '') for e in extra)}.so"
    srcs = [os.path.join(CS, f) for f in ("gkg_stem.hip", "gkg_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           f"-DSTEM_ABL={bits}", "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-o", so] + ["-D" + e for e in extra] + srcs)
    lib = C.CDLL(so)
    lib.gkg_stem_conv3x3s2_fwd.restype = C.c_int
    lib.gkg_stem_conv3x3s2_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]
    return lib


def main():
    torch.manual_seed(0)
    B, cin, H, W = 32, 3, 576, 576
    x = torch.randn(B, cin, H, W, device="cuda")
    res = {}
    for v in (sys.argv[1:] or list(VARIANTS)):
        name, *extra = v.split("+")
        lib = build(VARIANTS[name], tuple(extra))
        for cout in (40, 64):
            w = torch.randn(cout, cin, 3, 3, device="cuda")
            a = torch.rand(cout, device="cuda"); c = torch.randn(cout, device="cuda")
            for od, nm in ((0, "f32"), (1, "bf16")):
                out = torch.empty(B, H // 2, W // 2, cout, device="cuda", dtype=torch.float32 if od == 0 else torch.bfloat16)
                def call():
                    rc = lib.gkg_stem_conv3x3s2_fwd(x.data_ptr(), w.data_ptr(), None, a.data_ptr(), c.data_ptr(), out.data_ptr(), B, cin, H, W, cout, 1, od, None)
                    assert rc == 0, rc
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record(); e1.synchronize()
                res.setdefault(v, {})[f"{cout}/{nm}"] = round(e0.elapsed_time(e1) * 50, 1)
        print(v, res[v], flush=True)


if __name__ == "__main__":
    main()
