"""Framebuffer sharding across GPUs.  The scene is replicated; ONE frame is split by 8-row tile rows of the
reference grid (kernel.cpp:128-133): rank r of N renders the tile rows r, r + N, r + 2N, ... (vxrt_render_interleaved / DCR
0x7F3), with no data-path collective, and one gather assembles the image on rank 0.  Interleaving is what balances the ranks --
the cost of a tile varies 4x over a frame, mostly with image height, so contiguous bands (row_bands, kept for callers that
want them) finish at different times.  Strong scaling: the frame, and so the total work, is fixed as N grows."""
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


def interleaved_tile_rows(height, rank, world):
    """Tile rows of rank `rank`: rank, rank + world, ... below ceil(height / 8)."""
    return list(range(rank, (height + TILE - 1) // TILE, world))


def interleaved_rows(height, rank, world):
    """Frame rows (numpy int array) rank `rank` renders."""
    rows = [np.arange(t * TILE, min((t + 1) * TILE, height)) for t in interleaved_tile_rows(height, rank, world)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def padded_share_rows(height, world):
    """Rows of the equal-size buffer every rank contributes to the gather: ceil(tile rows / world) * 8."""
    tiles = (height + TILE - 1) // TILE
    return ((tiles + world - 1) // world) * TILE


def extract_interleaved(frame, height, rank, world):
    """The rows of rank `rank` out of its full-size frame buffer (torch tensor [height, width]) as one contiguous
    [padded_share_rows, width] tensor (rows past the rank's share are zero)."""
    import torch
    width = frame.shape[1]
    tiles = (height + TILE - 1) // TILE
    per = (tiles + world - 1) // world
    out = torch.zeros((per * TILE, width), dtype=frame.dtype, device=frame.device)
    if height % TILE == 0 and tiles % world == 0:
        # frame viewed as [per, world, 8, width]: this rank's share is one strided slice -> a single copy kernel
        out.view(per, TILE, width).copy_(frame.view(per, world, TILE, width)[:, rank])
        return out
    for k, t in enumerate(interleaved_tile_rows(height, rank, world)):
        y0, y1 = t * TILE, min((t + 1) * TILE, height)
        out[k * TILE: k * TILE + (y1 - y0)] = frame[y0:y1]
    return out


def assemble_interleaved(parts, height, width, world):
    """Full [height, width] frame from the `world` padded shares (list of [padded_share_rows, width] tensors, rank order)."""
    import torch
    tiles = (height + TILE - 1) // TILE
    per = (tiles + world - 1) // world
    if height % TILE == 0 and tiles % world == 0:
        stack = torch.stack([p.view(per, TILE, width) for p in parts], dim=1)     # [per, world, 8, width]
        return stack.reshape(height, width)
    frame = torch.empty((height, width), dtype=parts[0].dtype, device=parts[0].device)
    for r in range(world):
        for k, t in enumerate(interleaved_tile_rows(height, r, world)):
            y0, y1 = t * TILE, min((t + 1) * TILE, height)
            frame[y0:y1] = parts[r][k * TILE: k * TILE + (y1 - y0)]
    return frame


def gather_interleaved(share, height, width, rank, world, group=None):
    """share: this rank's padded share (extract_interleaved).  Returns the full frame on rank 0 (None elsewhere).  One
    collective: gather of equal-size shares (RCCL over xGMI with backend 'nccl', gloo on CPU tensors)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return assemble_interleaved([share], height, width, 1)
    out = [torch.empty_like(share) for _ in range(world)] if rank == 0 else None
    dist.gather(share, out, dst=0, group=group)
    if rank != 0:
        return None
    return assemble_interleaved(out, height, width, world)


def gather_frame(band, height, width, rank, world, group=None):
    """Contiguous-band variant: band = torch tensor [rows_of_this_rank, width] (row_bands).  Returns the full frame on rank 0."""
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


class InterleavedGather:
    """The per-frame image assembly of an N-rank run with everything that can be prepared once prepared once: frames are rendered
    into buffers of padded height (padded_share_rows * world rows, so that a rank's share is ONE strided slice whatever
    height % (8 * world) is), the share, the receive buffers and the assembled frame are preallocated per slot, and a frame costs
    one strided copy on every rank + one gather + one interleaving copy on rank 0 -- three launches, not one per tile row
    (1080 rows = 135 tile rows divide by no N > 1: the tile-row loop of extract/assemble_interleaved was the whole frame time).
    batch > 1: a slot holds `batch` frames ([batch, padded_height, width], what vxrt_render_interleaved_batch fills) and the same
    three launches move all of them."""

    def __init__(self, height, width, rank, world, device, slots=2, dtype=None, group=None, collective=True, batch=1,
                 single_rank_collective=False):
        import torch
        self.h, self.w, self.rank, self.world, self.group, self.batch = height, width, rank, world, group, batch
        self.per = padded_share_rows(height, world) // TILE        # tile rows per rank, padded
        self.padded_height = self.per * world * TILE
        self.collective = collective                                # False: rehearsal of one rank without the network
        # True: a group of ONE rank still goes through dist.gather (the RCCL call of the N-rank run, exercised on a 1-GPU box)
        self.single_rank_collective = single_rank_collective
        self.dtype = dtype or torch.int32
        self.shares = [torch.zeros((batch, self.per * TILE, width), dtype=self.dtype, device=device) for _ in range(slots)]
        self.recv = self.full = None
        if rank == 0:
            self.recv = [[torch.zeros((batch, self.per * TILE, width), dtype=self.dtype, device=device) for _ in range(world)] for _ in range(slots)]
            self.full = [torch.zeros((batch, self.per, world, TILE, width), dtype=self.dtype, device=device) for _ in range(slots)]

    def new_frame_buffer(self, device):
        """[padded_height, width] (batch 1) or [batch, padded_height, width]: what the render calls write this rank's rows into."""
        import torch
        shape = (self.padded_height, self.w) if self.batch == 1 else (self.batch, self.padded_height, self.w)
        return torch.zeros(shape, dtype=self.dtype, device=device)

    @property
    def frame_stride(self):
        """Pixels between two frames of a batch buffer (dst_frame_stride of vxrt_render_interleaved_batch)."""
        return self.padded_height * self.w

    def gather(self, frame, slot, via_cpu=False):
        """frame: buffer from new_frame_buffer this rank rendered its tile rows into.  Returns the assembled frame(s) on rank 0
        ([height, width] or [batch, height, width], views of a per-slot buffer), None elsewhere."""
        import torch
        import torch.distributed as dist
        share = self.shares[slot]
        share.view(self.batch, self.per, TILE, self.w).copy_(frame.view(self.batch, self.per, self.world, TILE, self.w)[:, :, self.rank])
        if self.collective and (self.world > 1 or self.single_rank_collective):
            if via_cpu:     # gloo rehearsal: collectives on host tensors
                out = [torch.empty_like(share, device="cpu") for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(share.cpu(), out, dst=0, group=self.group)
                if self.rank == 0:
                    for r in range(self.world):
                        self.recv[slot][r].copy_(out[r])
            else:
                dist.gather(share, self.recv[slot] if self.rank == 0 else None, dst=0, group=self.group)
        elif self.rank == 0:
            self.recv[slot][0].copy_(share)
        return self.assemble(slot)

    def assemble(self, slot):
        """Rank 0: the frame(s) from the shares received into `slot` (one interleaving copy)."""
        import torch
        if self.rank != 0:
            return None
        torch.stack([p.view(self.batch, self.per, TILE, self.w) for p in self.recv[slot]], dim=2, out=self.full[slot])
        out = self.full[slot].view(self.batch, self.padded_height, self.w)[:, : self.h]
        return out[0] if self.batch == 1 else out
