"""Frozen positional bias added to the k-NN distances (``Grapher.relative_pos``).

Same constant as the reference builds (pos_embed.py:21-85 -> torch_vertex.py:309-323):
``-(2 * PE @ PE.T / D)`` with PE the 2-D sin/cos embedding of the token grid, resized as an *image*
with bicubic interpolation to (n, n / r^2) when keys are pooled (r > 1).  Pinned by tests/golden/f9_relpos.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


def _axis_embedding(dim: int, coords: np.ndarray) -> np.ndarray:
    """[sin(p*w_i) | cos(p*w_i)], w_i = 10000^(-2i/dim), i < dim/2; float64 like the reference."""
    freq = 1.0 / np.power(10000.0, np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    phase = coords.reshape(-1, 1).astype(np.float64) * freq.reshape(1, -1)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def grid_embedding(embed_dim: int, grid: int) -> np.ndarray:
    """(grid*grid, embed_dim): first half encodes the column (w) index, second half the row (h) index —
    the reference's 'w goes first' meshgrid (pos_embed.py:44-47,56-63)."""
    cols, rows = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    return np.concatenate([_axis_embedding(embed_dim // 2, cols), _axis_embedding(embed_dim // 2, rows)], axis=1)


def relative_pos_matrix(embed_dim: int, grid: int) -> np.ndarray:
    pe = grid_embedding(embed_dim, grid)
    return 2.0 * (pe @ pe.T) / pe.shape[1]


_cache = {}


def build_relative_pos(in_channels: int, n: int, r: int) -> torch.Tensor:
    """(1, n, n // r^2) fp32, already negated (it is *added* to squared distances).

    Default: numpy float64 + CPU bicubic, bit-identical to what the reference stores in its checkpoints
    (pinned by tests/golden/f9_relpos).  At 576x576 inputs the first stage needs a 20736 x 20736 float64
    intermediate (3.4 GB, ~40 s on the host), so results are cached per (C, n, r) and ``GKG_RELPOS_DEVICE=cuda``
    evaluates the same formula on the GPU (float64 GEMM + bicubic resize; equal to ~1e-7, not bitwise) for
    random-init benchmarking.  Real checkpoints carry ``relative_pos`` themselves."""
    key = (in_channels, n, r, os.environ.get("GKG_RELPOS_DEVICE", "cpu"))
    if key not in _cache:
        grid = int(n ** 0.5)
        if key[3] != "cpu" and torch.cuda.is_available():
            pe = torch.from_numpy(grid_embedding(in_channels, grid)).to(key[3])
            base = (2.0 * (pe @ pe.T) / pe.shape[1]).to(torch.float32)[None, None]
            del pe
            out = F.interpolate(base, size=(n, n // (r * r)), mode="bicubic", align_corners=False)
            _cache[key] = (-out.squeeze(1)).cpu()
        else:
            base = torch.from_numpy(relative_pos_matrix(in_channels, grid).astype(np.float32))[None, None]
            base = F.interpolate(base, size=(n, n // (r * r)), mode="bicubic", align_corners=False)
            _cache[key] = -base.squeeze(1)
    return _cache[key].clone()


def resize_relative_pos(relative_pos: Optional[torch.Tensor], n_built: int, r: int, H: int, W: int):
    """Run-time re-interpolation when the feature map differs from the size the module was built for
    (reference Grapher._get_relative_pos, torch_vertex.py:317-323)."""
    if relative_pos is None or H * W == n_built:
        return relative_pos
    N = H * W
    return F.interpolate(relative_pos.unsqueeze(0), size=(N, N // (r * r)), mode="bicubic").squeeze(0)
