"""Framebuffer sharding across GPUs: the scene is replicated, rows are split into
contiguous bands aligned to the 8-row tile of the reference grid (kernel.cpp:128-133), each rank
renders its band with no data-path collective, and one gather assembles the image on rank 0."""
import numpy as np

TILE = 8


def row_bands(height, world):
    """[(y0, y1)] per rank: contiguous, tile-aligned, covering [0, height); sizes differ by <= 1 tile."""
    tiles = (height + TILE - 1) // TILE
    base, extra = divmod(tiles, world)
    bands, t = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        y0, y1 = min(t * TILE, height), min((t + n) * TILE, height)
        bands.append((y0, y1))
        t += n
    return bands


def max_band_rows(height, world):
    return max(y1 - y0 for y0, y1 in row_bands(height, world))


def gather_frame(band, height, width, rank, world, group=None):
    """band: torch int32 tensor [rows_of_this_rank, width] on this rank's device.  Returns the full
    [height, width] frame on rank 0 (None elsewhere).  One collective: gather of equal-size padded
    bands (RCCL over xGMI with backend 'nccl', gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return band
    rows = max_band_rows(height, world)
    pad = torch.zeros((rows, width), dtype=band.dtype, device=band.device)
    pad[: band.shape[0]] = band
    out = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, out, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.empty((height, width), dtype=band.dtype, device=band.device)
    for r, (y0, y1) in enumerate(row_bands(height, world)):
        frame[y0:y1] = out[r][: y1 - y0]
    return frame
