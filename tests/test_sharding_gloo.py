"""CPU, world_size 2 over gloo: the multi-GPU host logic (row bands / per-sample frames + one gather)
with the oracle standing in for the device kernel -- checks the assembly, not the traversal."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_bands_are_tile_aligned_and_cover_the_frame():
    import importlib
    sys.path.insert(0, ROOT)
    sh = importlib.import_module("vortex-raytracing_amd.sharding")
    for h in (1, 7, 8, 64, 135, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            b = sh.row_bands(h, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == h
            for (a0, a1), (b0, b1) in zip(b, b[1:]):
                assert a1 == b0
            assert all(y0 % 8 == 0 or y0 == h for y0, _ in b)
            rows = [y1 - y0 for y0, y1 in b]
            assert max(rows) - min(rows) < 16 or h < 8 * world   # one tile, plus a ragged last tile


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vrt = importlib.import_module("vortex-raytracing_amd")
    from oracle import pyoracle as po
    sc = vrt.scene.procedural("cornell")
    w, h = 40, 44
    if mode == "rows":
        y0, y1 = vrt.sharding.row_bands(h, world)[rank]
        px, _, _ = po.render(sc, w, h, y0=y0, y1=y1)      # stand-in for vxrt_render(y0, y1)
        band = torch.from_numpy(px[y0:y1].view(np.int32).copy())
        frame = vrt.sharding.gather_frame(band, h, w, rank, world)
        if rank == 0:
            q.put(frame.numpy().view(np.uint32))
    else:   # one sample (full frame) per rank; rank 0 keeps the last one, as kernel.cpp:67-80 overwrites
        px, _, _ = po.render(sc, w, h)
        t = torch.from_numpy(px.view(np.int32).copy())
        out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, out, dst=0)
        if rank == 0:
            assert all(torch.equal(o, out[0]) for o in out)
            q.put(out[-1].numpy().view(np.uint32))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rows", "samples"])
def test_two_rank_frame_assembly(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import importlib
    vrt = importlib.import_module("vortex-raytracing_amd")
    from oracle import pyoracle as po
    want, _, _ = po.render(vrt.scene.procedural("cornell"), 40, 44)
    assert np.array_equal(frame, want)
