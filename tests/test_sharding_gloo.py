"""World_size 2 over gloo: the multi-GPU host logic -- one frame split by interleaved 8-row tile rows (the default of bench.py
--gpus N) or by contiguous row bands, each rank rendering its share with no data-path collective, ONE gather assembling the
image on rank 0.

CPU tests (not marked gpu): the oracle stands in for the device kernel, so they check the split and the assembly.  The gpu-marked
test runs the same two ranks with the HIP path (vxrt_render_interleaved / vxrt_render on the one GPU of the box, shares gathered
through host memory over gloo) and compares the assembled frame with the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharding():
    import importlib
    sys.path.insert(0, ROOT)
    return importlib.import_module("vortex-raytracing_amd.sharding")


def test_row_bands_are_tile_aligned_and_cover_the_frame():
    sh = _sharding()
    for h in (1, 7, 8, 64, 135, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            b = sh.row_bands(h, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == h
            for (a0, a1), (b0, b1) in zip(b, b[1:]):
                assert a1 == b0
            assert all(y0 % 8 == 0 or y0 == h for y0, _ in b)
            rows = [y1 - y0 for y0, y1 in b]
            assert max(rows) - min(rows) < 16 or h < 8 * world   # one tile, plus a ragged last tile


def test_interleaved_tile_rows_partition_the_frame_and_reassemble():
    sh = _sharding()
    for h, w, world in ((1080, 24, 8), (1080, 24, 2), (2160, 8, 8), (1076, 16, 3), (45, 16, 4), (7, 8, 2), (64, 8, 16)):
        rows = np.concatenate([sh.interleaved_rows(h, r, world) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(h))
        for r in range(world):
            assert all((y // 8) % world == r for y in sh.interleaved_rows(h, r, world))
        f = torch.arange(h * w, dtype=torch.int32).reshape(h, w)
        parts = [sh.extract_interleaved(f, h, r, world) for r in range(world)]
        assert all(p.shape == (sh.padded_share_rows(h, world), w) for p in parts)
        assert torch.equal(sh.assemble_interleaved(parts, h, w, world), f)
    # balance: at 1080p over 8 ranks every rank gets 16 or 17 of the 135 tile rows
    n = [len(sh.interleaved_tile_rows(1080, r, 8)) for r in range(8)]
    assert max(n) - min(n) <= 1 and sum(n) == 135


def _worker(rank, world, port, mode, use_hip, q):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vrt = importlib.import_module("vortex-raytracing_amd")
    sh = vrt.sharding
    sc = vrt.scene.procedural("atrium", 3, 0, 3) if use_hip else vrt.scene.procedural("cornell")
    w, h = (328, 184) if use_hip else (40, 44)
    if use_hip:
        # the HIP path: this rank's share of the frame through the C ABI, into its own full-size frame buffer
        dev = "cuda:0"
        ds = vrt.tracer.DeviceScene(sc, dev)
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = (300.0, 480.0, 60.0)
        buf = torch.full((h, w), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        if mode == "tilerows":
            vrt.rtapi.render_interleaved(ds.accel, w, h, rank, world, p, buf.data_ptr(), 1, None, None, None, s)
        else:
            y0, y1 = sh.row_bands(h, world)[rank]
            vrt.rtapi.render(ds.accel, w, h, y0, y1, p, buf.data_ptr(), 1, None, None, None, s)
        torch.cuda.synchronize()
        assert vrt.rtapi.status(s) == 0
        frame_local = buf.cpu()
    else:
        from oracle import pyoracle as po
        px, _, _ = po.render(sc, w, h)                      # stand-in for the kernel: full frame, the share is cut out below
        frame_local = torch.from_numpy(px.view(np.int32).copy())
    if mode == "tilerows":
        share = sh.extract_interleaved(frame_local, h, rank, world)
        frame = sh.gather_interleaved(share, h, w, rank, world)
    else:
        y0, y1 = sh.row_bands(h, world)[rank]
        frame = sh.gather_frame(frame_local[y0:y1].contiguous(), h, w, rank, world)
    if rank == 0:
        q.put(frame.numpy().view(np.uint32))
    dist.destroy_process_group()


def _run(mode, use_hip):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, use_hip, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return frame


@pytest.mark.parametrize("mode", ["tilerows", "rows"])
def test_two_rank_frame_assembly(mode):
    frame = _run(mode, False)
    sys.path.insert(0, ROOT)
    import importlib
    vrt = importlib.import_module("vortex-raytracing_amd")
    from oracle import pyoracle as po
    want, _, _ = po.render(vrt.scene.procedural("cornell"), 40, 44)
    assert np.array_equal(frame, want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tilerows", "rows"])
def test_two_rank_frame_assembly_on_the_hip_path(mode, po):
    """Two processes share the box's GPU; each renders its share with the HIP kernels, gloo gathers, rank 0 assembles."""
    frame = _run(mode, True)
    sys.path.insert(0, ROOT)
    import importlib
    vrt = importlib.import_module("vortex-raytracing_amd")
    want, _, _, _ = po.render_ex(vrt.scene.procedural("atrium", 3, 0, 3), 328, 184, po.shade_params(light_pos=(300.0, 480.0, 60.0)), 1)
    assert np.array_equal(frame, want)
