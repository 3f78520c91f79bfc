#!/usr/bin/env python
"""HBM-side traffic of gemm_x6_kernel at the stage shapes against its algorithmic bytes (does a second column tile re-fetch the
activations from HBM?).  Two modes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/x6_fetch -o run -- python tools/ubench/x6_traffic.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/x6_write -o run -- python tools/ubench/x6_traffic.py run
    python tools/ubench/x6_traffic.py report /tmp/x6_fetch /tmp/x6_write
The counters are calibrated in the same run on the 512 MiB in-place add that evicts the caches between launches (reads and
writes 512 MiB with 16-byte-per-lane accesses; FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes
of such reads: MI355X guide)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [(663552, 80, 80), (663552, 80, 128), (663552, 320, 80), (663552, 80, 320), (165888, 160, 160), (41472, 400, 400)]
CAL_BYTES = (128 << 20) * 4             # the cache-eviction add: 512 MiB read, 512 MiB written


def run():
    import ctypes
    import torch
    from gkgnet_amd import _lib
    lib = _lib.load()
    flush = torch.empty(128 << 20, dtype=torch.float32, device="cuda")
    for _ in range(3):
        flush.add_(1.0)
    for R, cin, cout in SHAPES:
        x = torch.randn(R, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") * 0.1
        y = torch.empty(R, cout, device="cuda")
        pf = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, 1, 0), dtype=torch.uint8, device="cuda")
        host = ctypes.create_string_buffer(lib.gkg_x6_prep_desc_bytes())
        units = lib.gkg_x6_prep_desc_fill(host, 0, w.data_ptr(), pf.data_ptr(), None, cin, cout, 1, 0, 0)
        descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).cuda()
        _lib.check(lib.gkg_x6_prep_weights(descs.data_ptr(), 1, units, None), "prep")
        for _ in range(3):
            flush.add_(1.0)                  # evict the operands (L2 + memory-side cache)
            _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, 1, 0,
                                                *([None] * 10), 0.0, 0.0, None, None), "fwd")
        # the weight gradient of the same layer: dW = dY^T X over the R rows (reads both activations, writes a few KB)
        dy = torch.randn(R, cout, device="cuda")
        dw = torch.zeros(cout, cin, device="cuda")
        for _ in range(3):
            flush.add_(1.0)
            dw.zero_()
            _lib.check(lib.gkg_linear_wgrad_x6(dy.data_ptr(), cout, R * cout, x.data_ptr(), cin, R * cin, dw.data_ptr(), R, cin, cout,
                                               1, 0, None), "wgrad")
        torch.cuda.synchronize()


def load(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], int(r.get("Grid_Size", 0)), float(r["Counter_Value"])))
    return sorted(rows)


def report(dfetch, dwrite):
    out = {}
    for counter, d in (("FETCH_SIZE", dfetch), ("WRITE_SIZE", dwrite)):
        rows = load(d, counter)
        cal = [v for _, n, g, v in rows if "vectorized_elementwise_kernel" in n and g >= (128 << 20) // 8]
        if not cal:
            print(counter, "no calibration kernel found; kernels seen:", sorted({n[:60] for _, n, _, _ in rows})[:12]); continue
        k = CAL_BYTES / (sum(cal[-4:]) / len(cal[-4:]) * 1024)             # true bytes per reported byte
        print(f"{counter}: the 512 MiB add reports {sum(cal[-4:]) / len(cal[-4:]) / 1024:.1f} MiB -> factor {k:.3f}")
        x6 = [(n, g, v) for _, n, g, v in rows if "gemm_x6_kernel" in n]
        for i, (R, cin, cout) in enumerate(SHAPES):
            vals = [v for n, g, v in x6[3 * i:3 * i + 3]]
            if vals:
                out.setdefault((R, cin, cout), {})[counter] = vals[-1] * 1024 * k
        wg = [(n, g, v) for _, n, g, v in rows if "wgrad_x6" in n and "wgrad_x6_kernel" not in n]      # the main (DMA) launch of each call
        for i, (R, cin, cout) in enumerate(SHAPES):
            vals = [v for n, g, v in wg[3 * i:3 * i + 3]]
            if vals and counter == "FETCH_SIZE":
                print(f"   wgrad R={R:6d} {cin:3d}x{cout:3d}: fetched {vals[-1] * 1024 * k / 1e6:7.1f} MB, algorithmic {R * (cin + cout) * 4 / 1e6:7.1f} "
                      f"-> {vals[-1] * 1024 * k / (R * (cin + cout) * 4):.2f}")
    for (R, cin, cout), v in out.items():
        rd, wr = v.get("FETCH_SIZE", float("nan")), v.get("WRITE_SIZE", float("nan"))
        print(f"R={R:6d} {cin:3d}->{cout:3d}: fetched {rd / 1e6:7.1f} MB (algorithmic {R * cin * 4 / 1e6:6.1f} + planes), written {wr / 1e6:7.1f} MB "
              f"(algorithmic {R * cout * 4 / 1e6:6.1f}); total / algorithmic = {(rd + wr) / (R * (cin + cout) * 4):.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
