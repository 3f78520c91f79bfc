#!/usr/bin/env python
"""Where the buffered-selection k-NN launches of pvig_m (cfg5) go: compile-time ablations of knn_tile_kernel (-DKNN_ABLATE=n
builds of the k-NN sources, tools/ubench/_knn_ablate/) at the model's stage shapes.
   0  the shipped kernel
   1  every candidate is tested against a bound nothing passes (no appends, no flush work after the first)
   2  no selection at all: contraction, distance adds, one min per candidate
   3  the shipped kernel + counters: flushes, batches that took the merge network, admitted candidates, max-over-lanes sum
python tools/ubench/knn_ablate.py build   (here: cross-compiles)      python tools/ubench/knn_ablate.py   (on the GPU box)"""
import ctypes as C, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ubench", "_knn_ablate")
SRCS = ["gkg_api.hip", "gkg_knn.hip", "gkg_knn_f32.hip", "gkg_knn_f32_norp.hip", "gkg_knn_f32_mr.hip", "gkg_knn_f32_mr_norp.hip",
        "gkg_knn_bf.hip", "gkg_knn_bf_norp.hip", "gkg_knn_pf.hip", "gkg_knn_pf_norp.hip"]
ABLATED = {"gkg_knn.hip", "gkg_knn_f32.hip", "gkg_knn_f32_norp.hip"}
VARIANTS = (0, 1, 2, 3)


def build():
    os.makedirs(OUT, exist_ok=True)
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
            "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"), "-I" + CS]
    jobs = []
    for s in SRCS:
        if s in ABLATED:
            for v in VARIANTS:
                jobs.append((s, os.path.join(OUT, f"{s}.{v}.o"), [f"-DKNN_ABLATE={v}"]))
        else:
            jobs.append((s, os.path.join(OUT, f"{s}.o"), []))
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(lambda j: subprocess.check_call(["/opt/rocm/bin/hipcc"] + base + j[2] + ["-c", os.path.join(CS, j[0]), "-o", j[1]]), jobs))
    for v in VARIANTS:
        objs = [os.path.join(OUT, f"{s}.{v}.o" if s in ABLATED else f"{s}.o") for s in SRCS]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libknn_ablate{v}.so")] + objs)
    for f in os.listdir(OUT):
        if f.endswith(".o"):
            os.remove(os.path.join(OUT, f))


def load(v):
    lib = C.CDLL(os.path.join(OUT, f"libknn_ablate{v}.so"))
    lib.gkg_knn_workspace_bytes.restype = C.c_size_t
    lib.gkg_knn_workspace_bytes.argtypes = [C.c_int] * 7 + [C.c_uint]
    lib.gkg_knn_fwd.restype = C.c_int
    lib.gkg_knn_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gkg_debug_set_knn_timeline.argtypes = [C.c_void_p]
    return lib


# pvig_m at 768 x 768, B = 16, G = 8 (cfg5): stage 1 / 2 (pooled keys), stage 3 (dilation 2); pvig_s stage-1 for scale
SHAPES = [("pvig_m stage 1  c=12 kd=18", 128, 12, 36864, 2304, 18, 1, True),
          ("pvig_m stage 2  c=24 kd=18", 128, 24, 9216, 2304, 18, 1, True),
          ("pvig_m stage 3  c=48 kd=36", 128, 48, 2304, 2304, 18, 2, False)]


def main():
    import torch
    libs = {v: load(v) for v in VARIANTS}
    torch.manual_seed(0)
    for name, BG, c, N, M, k, dil, pooled in SHAPES:
        x = torch.randn(BG, c, N, device="cuda")
        y = torch.randn(BG, c, M, device="cuda") if pooled else None
        r = -torch.rand(N, M, device="cuda")
        flags = 1 | 64 | 16                # normalise, |relpos| <= 1, no prefilter
        idx = torch.empty(BG, N, k, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(4096, dtype=torch.int64, device="cuda")
        out = []
        for v in VARIANTS:
            lib = libs[v]
            nb = lib.gkg_knn_workspace_bytes(BG, c, N, M, k, dil, 0, flags)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            lib.gkg_debug_set_knn_timeline(cnt.data_ptr() if v == 3 else None)
            call = lambda: lib.gkg_knn_fwd(x.data_ptr(), None if y is None else y.data_ptr(), r.data_ptr(), idx.data_ptr(), None, BG, c,
                                           N, M, k, dil, 0, flags, ws.data_ptr(), nb, None)
            assert call() == 0
            torch.cuda.synchronize()
            cnt.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                assert call() == 0
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 3 * 1e3)
        # the shipped kernel without the positional bias (its loads: 64 query rows x M x 4 B per workgroup, the same rows for every
        # (image, group) problem)
        flags_n = 1 | 16
        lib = libs[0]
        nb = lib.gkg_knn_workspace_bytes(BG, c, N, M, k, dil, 0, flags_n)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        lib.gkg_debug_set_knn_timeline(None)
        call = lambda: lib.gkg_knn_fwd(x.data_ptr(), None if y is None else y.data_ptr(), None, idx.data_ptr(), None, BG, c,
                                       N, M, k, dil, 0, flags_n, ws.data_ptr(), nb, None)
        assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            assert call() == 0
        e1.record()
        torch.cuda.synchronize()
        norp = e0.elapsed_time(e1) / 3 * 1e3
        cn = (cnt[:4].cpu().double() / 3).tolist()
        waves = BG * ((N + 63) // 64) * (1 if "stage 1" in name else 4)
        print(f"{name}: prep + k-NN launch, us: shipped {out[0]:.0f} | tested, none admitted {out[1]:.0f} | no selection {out[2]:.0f} | "
              f"with counters {out[3]:.0f} | shipped, no positional bias {norp:.0f}")
        print(f"    per wave: {cn[0] / waves:.1f} flushes ({cn[1] / waves:.1f} through the merge network), {cn[2] / waves / 64:.1f} admitted candidates per lane, "
              f"sum over flushes of the fullest lane's count {cn[3] / waves:.1f}; {M / (1 if 'stage 1' in name else 4):.0f} keys per wave", flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        main()
