"""Strip partition of a frame over ranks — pure host logic, no GPU.

The image is cut into strips of 8 pixel rows (one 8x8 thread-group row of the reference's dispatch,
RayTraceMaster.cs:806-810) dealt round-robin: rank r owns group rows r, r+N, r+2N, ...  Sky rows cost
~1 ray per pixel and object rows up to numBounces, so interleaving balances the load.  Pixels keep
their global id.xy (the RNG depends on it: RayTraceShader.compute:78,434), so the union of the ranks'
strips is bit-identical to a single-GPU frame.  One gather at frame end moves the strips to rank 0.
"""
from __future__ import annotations

import numpy as np


def group_rows(height: int) -> int:
    return (height + 7) // 8


def n_strips(height: int, rank: int, world: int) -> int:
    g = group_rows(height)
    return (g - rank + world - 1) // world if rank < g else 0


def strip_row_ranges(height: int, rank: int, world: int):
    """[(y0, y1), ...] pixel-row ranges (row 0 = bottom) owned by `rank`."""
    return [(8 * g, min(8 * g + 8, height)) for g in range(rank, group_rows(height), world)]


def packed_rows(height: int, world: int) -> int:
    """Rows of the dense per-rank buffer: every rank pads to rank 0's strip count (gather needs equal sizes)."""
    return 8 * n_strips(height, 0, world)


def pack_rows_host(image: np.ndarray, rank: int, world: int) -> np.ndarray:
    """(H, W, 4) -> (packed_rows, W, 4): this rank's strips, strip-major; missing rows are zero."""
    h, w = image.shape[:2]
    out = np.zeros((packed_rows(h, world), w, 4), dtype=np.float32)
    for j, (y0, y1) in enumerate(strip_row_ranges(h, rank, world)):
        out[8 * j: 8 * j + (y1 - y0)] = image[y0:y1]
    return out


def unpack_rows_host(parts, width: int, height: int) -> np.ndarray:
    """Inverse of pack_rows_host over all ranks: parts[r] is rank r's dense buffer."""
    world = len(parts)
    img = np.zeros((height, width, 4), dtype=np.float32)
    for r, p in enumerate(parts):
        p = np.asarray(p, dtype=np.float32).reshape(-1, width, 4)
        for j, (y0, y1) in enumerate(strip_row_ranges(height, r, world)):
            img[y0:y1] = p[8 * j: 8 * j + (y1 - y0)]
    return img


def gather_to_root(dist, mine, rank: int, world: int):
    """The single frame-end collective: gather equal-sized dense buffers (torch tensors, CPU with gloo or
    device with RCCL) to rank 0.  Returns the list of parts on rank 0, None elsewhere."""
    import torch
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    return parts


def running_mean_alpha(samples, start: float = 0.0) -> float:
    """Alpha channel of `_converged` after AdditionShader blends with _Sample = s for s in `samples` (in order), starting from a fresh
    (zero) image: the fragment's alpha is a = 1 / (s + 1) itself and is blended like the colours (AS:39-41), so every pixel holds
    w <- a * a + w * (1 - a) in float32 — exactly the operations of k_blit_add.  What the root passes to urt_texture_unpack_rows_rgb."""
    w = np.float32(start)                                     # (`start`: the alpha the image held before these samples)
    one = np.float32(1.0)
    for s in samples:
        a = one / (np.float32(s) + one)
        w = np.float32(np.float32(a * a) + np.float32(w * np.float32(one - a)))
    return float(w)
