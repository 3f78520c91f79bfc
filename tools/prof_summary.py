#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): totals over the LAST `--window-ms` of
GPU activity (the timed + profiled steps of bench.py), optionally the raw launch sequence of the last step.
    python tools/prof_summary.py gpurun_out/prof/x_results.db [--window-ms 15] [--sequence 140]"""
import argparse, sqlite3, sys
from collections import defaultdict

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--window-ms", type=float, default=15.0)
    ap.add_argument("--sequence", type=int, default=0)
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    tend = rows[-1][2]
    sel = [r for r in rows if r[1] > tend - args.window_ms * 1e6]
    agg = defaultdict(lambda: [0.0, 0])
    for n, s, e, *_ in sel:
        agg[n][0] += (e - s) / 1e3
        agg[n][1] += 1
    tot = sum(v[0] for v in agg.values())
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{n[:120]:120s} {t:9.1f} us x{c:4d} avg {t / c:7.1f} {100 * t / tot:5.1f}%")
    print(f"busy {tot / 1e3:.2f} ms in the last {args.window_ms} ms")
    if args.sequence:
        prev = None
        for n, s, e, gx, wx in sel[-args.sequence:]:
            gap = 0.0 if prev is None else (s - prev) / 1e3
            print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {gx:>9}/{wx:<4} {n[:90]}")
            prev = e

if __name__ == "__main__":
    main()
