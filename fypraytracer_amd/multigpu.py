"""Tile split of one frame across the GPUs of a node + the single image gather.

The path shards by pixel: every rank renders a contiguous band of image rows
from a full replica of the scene; ReSTIR Part 1 is recomputed on `halo` rows either side so
Part 2's spatial reuse finds its neighbours locally; temporal reservoirs / accumulation stay
on the rank that owns the rows; the only data that crosses the fabric is the final RGBA8
image (one all-gather per frame, RCCL over xGMI when the backend is "nccl").
"""
from __future__ import annotations

import numpy as np


def band_rows(height: int, world: int, rank: int) -> tuple[int, int]:
    """Rows [begin, end) of `rank`: equal bands of ceil(H / world) rows (the last may be short)."""
    per = (height + world - 1) // world
    return min(height, rank * per), min(height, (rank + 1) * per)


def halo_rows(settings, technique: int, world: int) -> int:
    """Halo (rows) for ReSTIR: the spatial-neighbour radius as the kernels use it (uint8 cast,
    Renderer.cu:1897); 0 for the pure per-pixel techniques or a single GPU."""
    if world <= 1 or technique not in (7, 8) or not settings.use_spatial_reuse:
        return 0
    return int(settings.spatial_neighbor_radius) & 0xFF


def gather_image(band, height: int, width: int, world: int, dist):
    """all-gather the per-rank row bands into the full (H, W) uint32/int32 image.
    `band` is a torch tensor holding this rank's rows (rows_per * W elements, zero padded)."""
    import torch
    per = (height + world - 1) // world
    full = torch.empty(world * per * width, dtype=band.dtype, device=band.device)
    dist.all_gather_into_tensor(full, band.contiguous())
    return full[: height * width].view(height, width)


def stitch(bands, height: int, width: int):
    """Host-side equivalent of the gather for tests: list of (r0, r1, image_rows) -> (H, W)."""
    out = np.zeros((height, width), dtype=np.uint32)
    for r0, r1, img in bands:
        out[r0:r1] = img[r0:r1]
    return out
