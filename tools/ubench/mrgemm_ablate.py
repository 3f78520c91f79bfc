#!/usr/bin/env python
"""Compile-time phase ablation of the fused aggregation + projection kernel (csrc/gkg_mrgemm.hip): builds private copies of
the library with -DMG_ABL=<bits> into /tmp and times gkg_mr_linear_bf16 at the cfg3 stage shapes with HIP events.
    python tools/ubench/mrgemm_ablate.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
SHAPES = {"s1": (32, 2, 80, 20736, 1296, 9), "s2": (32, 2, 160, 5184, 1296, 9), "s3": (32, 2, 400, 1296, None, 9), "s4": (32, 2, 640, 324, None, 9)}
VARIANTS = {"full": 0, "no_gather": 1, "no_mfma": 2, "no_gelu": 4, "no_idx": 8, "no_store": 16, "no_gather_idx": 9, "skeleton": 31}


def build(bits):
    so = f"/tmp/libmg_{bits}.so"
    srcs = [os.path.join(CS, f) for f in (os.environ.get("MRGEMM_SRC", "gkg_mrgemm.hip"), "gkg_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           f"-DMG_ABL={bits}", "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-o", so] + srcs)
    lib = C.CDLL(so)
    lib.gkg_mr_linear_bf16.restype = C.c_int
    lib.gkg_mr_linear_bf16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 8 + [C.c_void_p]
    return lib


def main():
    torch.manual_seed(0)
    data = {}
    for name, (B, G, Cc, N, M, k) in SHAPES.items():
        x = torch.randn(B, N, Cc, device="cuda")
        src = None if M is None else torch.randn(B, M, Cc, device="cuda")
        Mk = N if M is None else M
        # spatially local neighbours, like a real graph
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-20, 21, (B * G, N, k), device="cuda")) % Mk).contiguous()
        ci = Cc // 2
        ci_pad, co_pad = (ci + 15) // 16 * 16, (ci + 31) // 32 * 32
        planes = torch.randn(4, ci_pad // 8, co_pad, 8, device="cuda").bfloat16().contiguous()
        a = torch.rand(2 * Cc, device="cuda") + 0.5
        c = torch.randn(2 * Cc, device="cuda")
        out = torch.empty(B * N, 2 * Cc, dtype=torch.bfloat16, device="cuda")
        data[name] = (x, src, idx, planes, a, c, out, B, G, Cc // G, N, Mk, k)
    res = {n: {} for n in SHAPES}
    for vname, bits in VARIANTS.items():
        if len(sys.argv) > 1 and vname not in sys.argv[1:]:
            continue
        lib = build(bits)
        for name, (x, src, idx, planes, a, c, out, B, G, cg, N, Mk, k) in data.items():
            def call():
                rc = lib.gkg_mr_linear_bf16(x.data_ptr(), None if src is None else src.data_ptr(), idx.data_ptr(), planes.data_ptr(),
                                            a.data_ptr(), c.data_ptr(), out.data_ptr(), 2 * cg * G, B, G, cg, N, Mk, k, 1, None)
                assert rc == 0, rc
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record(); e1.synchronize()
            res[name][vname] = round(e0.elapsed_time(e1) * 50, 1)
    for n, r in res.items():
        print(n, r, flush=True)


if __name__ == "__main__":
    main()
