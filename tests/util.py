"""Shared helpers for the parity tests: fixture loading, the near-tie index protocol, keyed weights."""
from __future__ import annotations

import json
import os
import zlib

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Two neighbour ranks whose reference distances differ by less than this may legitimately swap
# between implementations that accumulate the fp32 dot product in a different order
# (SURVEY.md §7 "Bit-exact indices"; distances on normalised tokens are O(1), 1 ulp ~ 1.2e-7).
NEAR_TIE = 2e-6


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrays = {k: z[k] for k in z.files if k != "meta"}
    return meta, arrays


def state_from(arrays, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in arrays.items() if k.startswith(prefix)}


def grads_from(arrays):
    return {k[len("grad/"):]: torch.from_numpy(np.array(v)) for k, v in arrays.items() if k.startswith("grad/")}


def check_indices(mine, ref_idx, topd, topi, dilation=1, tol=NEAR_TIE):
    """Near-tie protocol.  mine / ref_idx: (BG,N,k) neighbour indices (kept ranks 0,d,2d,..).
    topd/topi: the reference's sorted top-(k*d+1) distances / indices.
    Every position must match exactly, except where the reference's own distance at that rank is
    within ``tol`` of the distance of the key we returned (an fp32-rounding swap).
    Returns the number of such tolerated swaps."""
    mine = np.asarray(mine).astype(np.int64)
    ref_idx = np.asarray(ref_idx).astype(np.int64)
    assert mine.shape == ref_idx.shape, (mine.shape, ref_idx.shape)
    bad = np.argwhere(mine != ref_idx)
    swaps = 0
    for b, n, j in bad:
        r = j * dilation
        cand = np.nonzero(topi[b, n] == mine[b, n, j])[0]
        assert cand.size == 1, f"(bg={b}, n={n}, rank={r}): key {mine[b, n, j]} not in the reference top-(kd+1)"
        gap = abs(float(topd[b, n, cand[0]]) - float(topd[b, n, r]))
        assert gap <= tol, f"(bg={b}, n={n}, rank={r}): wrong neighbour, reference gap {gap:.3e} > {tol}"
        swaps += 1
    return swaps


def keyed_fill_(state_dict, seed=0):
    """Same rule as tools/gen_golden.py::keyed_fill_ (crc32(key)-seeded deterministic weights)."""
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if key.endswith("num_batches_tracked") or key.endswith("relative_pos"):
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif ".1.weight" in key or key.endswith("bn.weight"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith(".bias"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() >= 2:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        t.copy_(v.to(t.dtype))


# ---- the grouped projection's operand buffer (include/gkg_hip.h "XM layout") ------------------------------------------------
def xm_pack(x, m):
    """x, m (T, C) -> XM (T, 2C): row = [x_0 | m_0 | x_1 | m_1 | x_2 | m_2 | x_3 | m_3], chunks of C / 4."""
    T, C = x.shape
    return torch.stack([x.reshape(T, 4, C // 4), m.reshape(T, 4, C // 4)], dim=2).reshape(T, 2 * C)


def xm_split(XM):
    """XM (T, 2C) -> (x, m), each (T, C)."""
    T, C2 = XM.shape
    v = XM.reshape(T, 4, 2, C2 // 8)
    return v[:, :, 0].reshape(T, C2 // 2), v[:, :, 1].reshape(T, C2 // 2)


def xm_interleaved(XM):
    """XM (T, 2C) -> the reference's interleaved channel order [x_0, m_0, x_1, m_1, ...] (T, 2C) (torch_vertex.py:57-61)."""
    x, m = xm_split(XM)
    return torch.stack([x, m], dim=2).reshape(XM.shape[0], -1)
