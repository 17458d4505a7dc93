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
    """The image assembly of an N-rank run with everything that can be prepared once prepared once: frames are rendered into buffers
    of padded height (padded_share_rows * world rows, so that a rank's share is ONE strided slice whatever height % (8 * world) is),
    the share, the receive buffers and the assembled frames are preallocated per slot, and a set of frames costs one packing launch
    on every rank + one gather + one unpacking launch on rank 0 (1080 rows = 135 tile rows divide by no N > 1: a loop over tile
    rows was the whole frame time).  batch > 1: a slot holds up to `batch` frames ([batch, padded_height, width], what
    vxrt_render_interleaved_batch fills); gather(..., k) moves the first k of them -- the sets of a run need not be equally large
    (bench.py tapers the last sets so that what nothing overlaps any more, the last set's gather, is small).

    wire = "rgb24" (default where the width is a multiple of 4): a share travels as 3 bytes per pixel -- pixels are 0x00RRGGBB
    (common.h:149-154), the top byte is always zero -- packed and expanded by vxrt_wire_pack / vxrt_wire_unpack (HIP; on CPU tensors,
    the 2-rank gloo tests, the same layout through torch indexing): a quarter less on the link, which is what the last set's gather
    leaves exposed.  wire = "rgba32": the 4-byte pixels as they are, strided torch copies (round 4's path)."""

    def __init__(self, height, width, rank, world, device, slots=2, dtype=None, group=None, collective=True, batch=1,
                 single_rank_collective=False, wire="auto"):
        import torch
        self.h, self.w, self.rank, self.world, self.group, self.batch = height, width, rank, world, group, batch
        self.per = padded_share_rows(height, world) // TILE        # tile rows per rank, padded
        self.padded_height = self.per * world * TILE
        self.collective = collective                                # False: rehearsal of one rank without the network
        # True: a group of ONE rank still goes through dist.gather (the RCCL call of the N-rank run, exercised on a 1-GPU box)
        self.single_rank_collective = single_rank_collective
        self.dtype = dtype or torch.int32
        if wire == "auto":
            wire = "rgb24" if (width % 4 == 0 and self.dtype == torch.int32) else "rgba32"
        if wire not in ("rgb24", "rgba32") or (wire == "rgb24" and (width % 4 != 0 or self.dtype != torch.int32)):
            raise ValueError("wire format %r needs int32 pixels and a width that is a multiple of 4" % (wire,))
        self.wire = wire
        self.device = device
        rows = self.per * TILE
        if wire == "rgb24":
            self.shares = [torch.zeros((batch, rows, width, 3), dtype=torch.uint8, device=device) for _ in range(slots)]
        else:
            self.shares = [torch.zeros((batch, rows, width), dtype=self.dtype, device=device) for _ in range(slots)]
        self.recv = self.full = None
        if rank == 0:
            # one tensor per slot, rank-major, frame-major inside a rank's part for EVERY set size k: the k frames of rank r are recv[slot][r][:k]
            self.recv = [torch.zeros((world,) + tuple(self.shares[0].shape), dtype=self.shares[0].dtype, device=device) for _ in range(slots)]
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

    def wire_bytes(self, k=None):
        """Bytes one rank puts on the link for a set of k frames."""
        k = self.batch if k is None else k
        return k * self.per * TILE * self.w * (3 if self.wire == "rgb24" else 4)

    # -- the two ends of the wire ------------------------------------------------------------------------------------------
    def _pack(self, frame, share, k):
        import torch
        fr = frame.view(self.batch, self.padded_height, self.w)
        if self.wire == "rgb24":
            if frame.is_cuda:
                from . import rtapi
                rtapi.wire_pack(fr.data_ptr(), self.frame_stride, self.w, self.per, self.world, self.rank, k, share.data_ptr(),
                                torch.cuda.current_stream(frame.device).cuda_stream)
            else:   # host tensors (gloo tests): the same bytes through torch indexing
                src = fr[:k].view(k, self.per, self.world, TILE, self.w)[:, :, self.rank].contiguous()
                share[:k].view(k, self.per, TILE, self.w, 3).copy_(src.view(torch.uint8).view(k, self.per, TILE, self.w, 4)[..., :3])
        else:
            share[:k].view(k, self.per, TILE, self.w).copy_(fr[:k].view(k, self.per, self.world, TILE, self.w)[:, :, self.rank])

    def _unpack(self, slot, k):
        import torch
        recv, full = self.recv[slot], self.full[slot]
        if self.wire == "rgb24":
            if recv.is_cuda:
                from . import rtapi
                rtapi.wire_unpack(recv.data_ptr(), recv.stride(0), self.w, self.per, self.world, k, full.data_ptr(), self.frame_stride,
                                  torch.cuda.current_stream(recv.device).cuda_stream)
            else:
                out = full[:k].view(torch.uint8).view(k, self.per, self.world, TILE, self.w, 4)
                out[..., :3] = recv[:, :k].view(self.world, k, self.per, TILE, self.w, 3).permute(1, 2, 0, 3, 4, 5)
                out[..., 3] = 0
        else:
            torch.stack([recv[r, :k].view(k, self.per, TILE, self.w) for r in range(self.world)], dim=2, out=full[:k])

    def gather(self, frame, slot, via_cpu=False, k=None):
        """frame: buffer from new_frame_buffer this rank rendered its tile rows of the first k frames into (k = batch by default).  Returns
        the assembled frame(s) on rank 0 ([height, width] for batch 1, else [k, height, width]: views of a per-slot buffer), None elsewhere."""
        import torch
        import torch.distributed as dist
        k = self.batch if k is None else int(k)
        if not 1 <= k <= self.batch:
            raise ValueError("a set of %d frames in a slot of %d" % (k, self.batch))
        share = self.shares[slot]
        self._pack(frame, share, k)
        if self.collective and (self.world > 1 or self.single_rank_collective):
            if via_cpu:     # gloo rehearsal: collectives on host tensors
                out = [torch.empty_like(share[:k], device="cpu") for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(share[:k].cpu(), out, dst=0, group=self.group)
                if self.rank == 0:
                    for r in range(self.world):
                        self.recv[slot][r, :k].copy_(out[r])
            else:
                dist.gather(share[:k], [self.recv[slot][r, :k] for r in range(self.world)] if self.rank == 0 else None, dst=0, group=self.group)
        elif self.rank == 0:
            self.recv[slot][0, :k].copy_(share[:k])
        return self.assemble(slot, k)

    def assemble(self, slot, k=None):
        """Rank 0: the frame(s) from the shares received into `slot` (one launch)."""
        if self.rank != 0:
            return None
        k = self.batch if k is None else int(k)
        self._unpack(slot, k)
        out = self.full[slot].view(self.batch, self.padded_height, self.w)[:k, : self.h]
        return out[0] if self.batch == 1 else out


# ---------------------------------------------------------------------------------------------------------------------------
# Contiguous bands balanced by cost (bench.py --shard bands, the default with several GPUs).
#
# Why bands after all: with interleaved tile rows every rank's share is scattered over the image, so rank 0 needs one strided
# extraction on every rank + one interleaving copy of the WHOLE batch of frames on rank 0 before an image exists -- measured
# (profiles/r04_*): 65 us per 10 frames on rank 0 of 8, a tenth of its run, and serial after the last traversal.  A contiguous band
# lies in the final image as one block: rank 0 receives every band straight into its place (grouped ncclSend / ncclRecv = a gather with
# per-rank sizes), renders its own band in place, and no copy kernel runs at all.  What interleaving bought -- balance: the cost
# of a tile varies 4x over the frame -- comes from the band HEIGHTS instead: plan_bands() cuts the rows at equal measured cost
# and bench.py refines the cut with the ranks' own frame times during the untimed settle phase (rebalance_bands).
# ---------------------------------------------------------------------------------------------------------------------------

def equal_bands(height, world, align=TILE):
    """[b_0 = 0, b_1, ..., b_world = height]: `world` contiguous bands of (nearly) equal height, cuts on multiples of `align`."""
    units = (height + align - 1) // align
    base, extra = divmod(units, world)
    b, t = [0], 0
    for r in range(world):
        t += base + (1 if r < extra else 0)
        b.append(min(t * align, height))
    b[-1] = height
    return b


def rebalance_bands(bounds, times, height, align=TILE, damping=1.0, min_rows=TILE):
    """New band boundaries from the time each rank needed for its band: the cost of a row is taken as constant inside a band
    (time / rows), the cumulative cost is cut into equal parts, cuts are rounded to multiples of `align` and every band keeps at
    least `min_rows` rows.  damping < 1 moves the cuts only part of the way (the measured time of a band includes what does not
    scale with its rows).  Deterministic in its inputs: every rank computes the same plan from the gathered times."""
    world = len(bounds) - 1
    rows = [bounds[r + 1] - bounds[r] for r in range(world)]
    if world == 1 or min(rows) <= 0 or min(times) <= 0:
        return list(bounds)
    dens = [times[r] / rows[r] for r in range(world)]          # cost per row inside band r
    cum = [0.0]
    for r in range(world):
        cum.append(cum[-1] + times[r])
    total = cum[-1]
    new = [0]
    for k in range(1, world):
        target = total * k / world
        r = max(i for i in range(world) if cum[i] <= target)   # band the k-th cut falls into
        y = bounds[r] + (target - cum[r]) / dens[r]
        y = bounds[k] + damping * (y - bounds[k])
        new.append(int(round(y / align)) * align)
    new.append(height)
    for k in range(1, world):                                  # monotone, every band at least min_rows
        new[k] = max(new[k], new[k - 1] + min_rows)
    for k in range(world - 1, 0, -1):
        new[k] = min(new[k], new[k + 1] - min_rows)
    return new


def auto_sets(steps, world, max_set=16):
    """Sizes of the sets of frames an N-rank run issues its `steps` timed frames in (bench.py --sets auto, tile rows interleaved over the
    ranks).  Every set's gather travels while the next set is traced, except the LAST one's, which the end of the run exposes: the last set
    is `world` frames -- a rank's share of a frame is 1/world of it, so that is ONE frame's bytes per link whatever the world size -- and
    the steps before it go out in equal sets of at most `max_set`.  Runs too short for that (steps < 2 * world) and single ranks get
    equal sets.  sum == steps; every size in 1..max(max_set, world)."""
    steps, world = int(steps), int(world)
    if world > 1 and steps >= 2 * world:
        head = steps - world
        n_head = -(-head // max_set)
        per = -(-head // n_head)
        return [min(per, head - g) for g in range(0, head, per)] + [world]
    n_groups = max(2 if steps > 1 else 1, -(-steps // max_set))
    per = -(-steps // n_groups)
    return [min(per, steps - g) for g in range(0, steps, per)]


def taper(steps, first_max=16, last=1, ratio=0.5):
    """Sizes of the sets of frames a run of `steps` frames is issued in when every set ends with a collective: large sets first
    (a rank's share of a frame is small against its GPU, many frames per launch keep it full), geometrically smaller ones towards
    the end, so that what nothing overlaps any more -- the shading and the gather of the LAST set -- is small.  sum == steps."""
    out, left = [], int(steps)
    while left > 0:
        s = left if left <= last else min(first_max, max(last, int(round(left * ratio))))
        out.append(s)
        left -= s
    return out


class BandGather:
    """Image assembly of an N-rank run with contiguous bands: rank r renders rows [bounds[r], bounds[r+1]) of every frame of a set.
    Rank 0 owns the final images ([frames, height, width] per slot) and renders its own band IN PLACE; every other rank renders
    into a compact [frames, rows_r, width] buffer (render target = buffer - bounds[r] * width, see vxrt_render_rows_batch) and
    sends frame f's band with one ncclSend; rank 0 posts the matching ncclRecv straight into final[f, b_r:b_{r+1}] -- all of a
    set's transfers in ONE group call (a gather with per-rank sizes).  No extraction, no staging, no interleaving copy.
    collective=False: rehearsal of one rank without the network (nothing is moved: rank 0's own band is already in place)."""

    def __init__(self, height, width, rank, world, bounds, device, set_sizes, dtype=None, group=None, collective=True, via_cpu=False):
        import torch
        self.h, self.w, self.rank, self.world, self.group = height, width, rank, world, group
        self.bounds = list(bounds)
        self.y0, self.y1 = self.bounds[rank], self.bounds[rank + 1]
        self.collective, self.via_cpu = collective, via_cpu
        self.dtype = dtype or torch.int32
        self.bufs = []
        for k in set_sizes:       # one buffer per slot, sized for the largest set that uses it
            shape = (k, height, width) if rank == 0 else (k, self.y1 - self.y0, width)
            self.bufs.append(torch.zeros(shape, dtype=self.dtype, device=device))

    def target(self, slot):
        """(dst pointer, frame stride in pixels) for vxrt_render_rows_batch(y0 = self.y0, y1 = self.y1) into slot `slot`."""
        b = self.bufs[slot]
        if self.rank == 0:
            return b.data_ptr(), self.h * self.w
        return b.data_ptr() - b.element_size() * self.y0 * self.w, (self.y1 - self.y0) * self.w

    def gather(self, slot, k):
        """Assemble the first k frames of slot `slot` on rank 0 (called on the stream the collective is to be ordered on).
        Returns the [k, height, width] images on rank 0, None elsewhere."""
        import torch
        import torch.distributed as dist
        buf = self.bufs[slot]
        if self.collective and self.world > 1:
            if self.via_cpu:      # gloo rehearsal: host tensors, placed the same way
                if self.rank == 0:
                    stage = [[torch.empty((self.bounds[r + 1] - self.bounds[r], self.w), dtype=self.dtype) for _ in range(k)] for r in range(1, self.world)]
                    reqs = [dist.irecv(stage[r - 1][f], src=r, group=self.group, tag=f) for r in range(1, self.world) for f in range(k)]
                    for q in reqs:
                        q.wait()
                    for r in range(1, self.world):
                        for f in range(k):
                            buf[f, self.bounds[r]:self.bounds[r + 1]].copy_(stage[r - 1][f])
                else:
                    host = buf[:k].cpu()
                    for q in [dist.isend(host[f], dst=0, group=self.group, tag=f) for f in range(k)]:
                        q.wait()
            else:
                if self.rank == 0:
                    ops = [dist.P2POp(dist.irecv, buf[f, self.bounds[r]:self.bounds[r + 1]], r, self.group)
                           for r in range(1, self.world) for f in range(k) if self.bounds[r + 1] > self.bounds[r]]
                else:
                    ops = [dist.P2POp(dist.isend, buf[f], 0, self.group) for f in range(k)] if self.y1 > self.y0 else []
                if ops:
                    for q in dist.batch_isend_irecv(ops):
                        q.wait()          # (NCCL: orders the current stream behind the transfer; does not block the host)
        return buf[:k] if self.rank == 0 else None
