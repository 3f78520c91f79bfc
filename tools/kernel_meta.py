#!/usr/bin/env python
"""Per-kernel resource metadata (scratch bytes, VGPR spills, registers, LDS) of the gfx950 code objects embedded in a
hipcc output (object file, device-only bundle or libgkg_hip.so).

    python tools/kernel_meta.py [gkgnet_amd/libgkg_hip.so] [name-filter-regex]

The file's `__CLANG_OFFLOAD_BUNDLE__` sections are parsed directly (no roc-obj tooling needed), every gfx950 ELF is handed
to `llvm-readelf --notes`, and the AMDGPU metadata note is read.  Used by tests/test_abi.py to assert that the max-relative
kernels carry no scratch memory."""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path: str, arch: str = "gfx950"):
    """Yields the bytes of every embedded device ELF for ``arch``."""
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and MAGIC not in data:
        yield data
        return
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if arch in triple and size:
                yield data[p + off:p + off + size]


def kernels(path: str):
    """{kernel symbol: {private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count, vgpr_count, group_segment_fixed_size}}"""
    out = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as fh:
            fh.write(blob)
            tmp = fh.name
        try:
            txt = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(tmp)
        cur = {}
        for line in txt.splitlines():
            s = line.strip()
            if s.startswith("- .agpr_count") or s.startswith("- .args"):        # a new kernel entry starts
                if cur.get("name"):
                    out[cur["name"]] = cur
                cur = {}
            mm = re.match(r"-?\s*\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|"
                          r"group_segment_fixed_size|agpr_count):\s*(\S+)", s)
            if mm:
                k, v = mm.group(1), mm.group(2)
                if k == "name":
                    if "name" not in cur or cur.get("_in_args"):
                        pass
                    cur["name_candidate"] = v
                else:
                    cur[k] = int(v)
            if s.startswith(".symbol:"):
                cur["name"] = s.split(":", 1)[1].strip()
        if cur.get("name"):
            out[cur["name"]] = cur
    for v in out.values():
        v.pop("name_candidate", None)
    return out


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return list(names)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "gkgnet_amd", "libgkg_hip.so")
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    ks = kernels(path)
    names = sorted(ks)
    for sym, dem in zip(names, demangle([n.replace(".kd", "") for n in names])):
        if pat and not pat.search(dem):
            continue
        k = ks[sym]
        print(f"scratch {k.get('private_segment_fixed_size', 0):5d}  vspill {k.get('vgpr_spill_count', 0):4d}  vgpr {k.get('vgpr_count', 0):4d}  "
              f"agpr {k.get('agpr_count', 0):4d}  lds {k.get('group_segment_fixed_size', 0):6d}  {dem[:150]}")
