#!/usr/bin/env python
"""Wave-level phase timeline of knn_tile_kernel (s_memtime stamps, -DKNN_TIMELINE build of the k-NN sources in /tmp):
where a workgroup's life goes at the cfg2 shapes.  python tools/ubench/knn_timeline.py"""
import ctypes as C, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import numpy as np

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
SRCS = ["gkg_api.hip", "gkg_knn.hip", "gkg_knn_f32.hip", "gkg_knn_f32_norp.hip", "gkg_knn_f32_mr.hip", "gkg_knn_f32_mr_norp.hip", "gkg_knn_bf.hip", "gkg_knn_bf_norp.hip",
        "gkg_knn_pf.hip", "gkg_knn_pf_norp.hip"]


def build():
    so = "/tmp/libgkg_knn_tl.so"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-Wno-pass-failed", "-DKNN_TIMELINE", "-I" + os.path.join(ROOT, "include"), "-I" + CS]
    objs = [f"/tmp/tl_{s}.o" for s in SRCS]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda so_: subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CS, so_[0]), "-o", so_[1]]),
                    zip(SRCS, objs)))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    lib = C.CDLL(so)
    lib.gkg_knn_workspace_bytes.restype = C.c_size_t
    lib.gkg_knn_workspace_bytes.argtypes = [C.c_int] * 7 + [C.c_uint]
    lib.gkg_knn_fwd.restype = C.c_int
    lib.gkg_knn_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gkg_debug_set_knn_timeline.argtypes = [C.c_void_p]
    return lib


def placement(g, name):
    """every workgroup's start / end stamp and (XCC, SE, CU): how many share a CU, and how the launch is spread in time"""
    g0 = g
    g = g[g[:, 0] > 0]
    hw, xcc = g[:, 2], g[:, 3] & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"== {name}: {len(g)} workgroups on {len(uniq)} CUs; workgroups per CU histogram {dict(zip(*np.unique(cnt, return_counts=True)))}")
    ids = np.nonzero(g0[:, 0] > 0)[0]
    x0 = xcc == 0
    print("   XCC 0, per CU: (workgroup id, start - first start on that CU, life)")
    for kcu in np.unique(key[x0]):
        sel = key == kcu
        st0 = g[sel, 0].min()
        print("     CU", int(kcu) & 0xff, [(int(i), int(a - st0), int(b - a)) for i, a, b in zip(ids[sel], g[sel, 0], g[sel, 1])])
    for x in range(8):
        m = xcc == x
        if not m.any():
            continue
        st, en = g[m, 0], g[m, 1]
        t0 = st.min()
        print(f"   XCC {x}: {m.sum()} workgroups; starts span {st.max() - t0} cycles; life min/median/max {np.min(en - st)}/{int(np.median(en - st))}/{np.max(en - st)}; "
              f"last end {en.max() - t0}")


def main():
    lib = build()
    torch.manual_seed(0)
    flags0 = 1
    for name, BG, c, N, M, rp in (("cfg2 grapher", 128, 80, 324, 324, True), ("cfg2 label", 128, 80, 80, 324, False),
                                  ("pvig_s stage 1 (prefilter kernel)", 64, 40, 20736, 1296, "xy"), ("pvig_s stage 3 d=2 (prefilter kernel)", 64, 200, 1296, 1296, "d2")):
        x = torch.randn(BG, c, N, device="cuda")
        y = None if rp in (True, "d2") else torch.randn(BG, c, M, device="cuda")
        r = -torch.rand(N, M, device="cuda") if rp else None
        dil = 2 if rp == "d2" else 1
        flags = flags0 | (64 if rp else 0)
        nb = lib.gkg_knn_workspace_bytes(BG, c, N, M, 9, dil, 0, flags)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        idx = torch.empty(BG, N, 9, dtype=torch.int64, device="cuda")
        tl = torch.zeros(24 * 8 * 32 + 4096 * 4, dtype=torch.int64, device="cuda")
        lib.gkg_debug_set_knn_timeline(tl.data_ptr())
        for it in range(4):
            tl.zero_()
            torch.cuda.synchronize()
            rc = lib.gkg_knn_fwd(x.data_ptr(), None if y is None else y.data_ptr(), None if r is None else r.data_ptr(), idx.data_ptr(),
                                 None, BG, c, N, M, 9, dil, 0, flags, ws.data_ptr(), nb, None)
            assert rc == 0
            torch.cuda.synchronize()
        full = tl.cpu().numpy()
        t = full[:24 * 8 * 32].reshape(24, 8, 32)
        placement(full[24 * 8 * 32:].reshape(4096, 4), name)
        t0 = t[t > 0].min()
        print(f"== {name}: BG={BG} c={c} N={N} M={M}; stamps in us after the launch's first stamp (100 MHz s_memtime assumed: /100)")
        print("   phases: 0 start, 1 queries staged, 2 barrier, 3/4 5/6 7/8.. tile contraction / selection done, 28 loop done, 29 barrier, 30 lists in LDS, 31 end")
        print("   (prefilter kernel: 2 staged, 28 stream done, 29 barrier, 30 merge + survivors done, 25 barrier, 26 exact pairs done, 27 barrier, 31 end)")
        for wg in range(24):
            for w in range(8):
                row = t[wg, w]
                if row[0] == 0:
                    continue
                s = " ".join(f"{p}:{int(row[p] - row[0])}" for p in sorted(range(32), key=lambda q: row[q]) if row[p] > 0)
                print(f"   wg {wg // 3 * 97 + wg % 3:4d} wave {w}: {s}")


if __name__ == "__main__":
    main()
