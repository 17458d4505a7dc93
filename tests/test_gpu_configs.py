"""GPU parity at BASELINE.json's FULL sizes: the frames bench.py and tools/config_bench.py time, compared with the oracle over
WHOLE frames.  The checker runs from a thread pool (oracle/pyoracle.py: render_ex_mt and friends -- the foreign calls release the
GIL, row ranges are disjoint), so a 1080p primary + shadow frame of the 1,048,576-triangle scene costs it about a second on the
box's 16 cores, the 3840x2160 frame and the 10M-triangle hairball's 25.5 M rays a few.

  configs[2] headline  1,048,576-triangle atrium, 1920x1080, primary + shadow, 4 frames in flight  -> every pixel, hit record, colour
  the timed call       vxrt_render_batch, 5 frames per set, sets alternating on two streams       -> every pixel of two frames
  a rank's batches     interleaved tile rows with the learned tile order                          -> every row of the share
  configs[2] as worded same scene, primary + one diffuse bounce                                     -> every pixel vs render_gi
  configs[1]           bunny-class blob framed to fill the view, 1024x1024, primary + shadow        -> every pixel
  configs[4]           10M-triangle hairball framed to fill the view, 1920x1080, 16 spp AO          -> every pixel, count, colour
  configs[3]           the atrium at 3840x2160 as N interleaved tile-row sets (the 8-GPU split of bench.py), assembled
                       with sharding.assemble_interleaved, equals the one-shot frame and the oracle's: every pixel
  software twin        the 1,048,576-triangle BVH2 at 1920x1080, frames alternating on two streams                  -> every pixel of every frame
  random-ray leg       bench.py's 16 Mi random rays through vxrt_trace in one launch                                -> every 16th hit record (1 Mi), bit for bit
  several GPUs         the headline frame through vx_start on three shares behind one vx_device (VORTEX_HIP_DEVICES) -> every pixel"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COLOR_RTOL = 1e-5
LIGHT = (300.0, 480.0, 60.0)     # bench.py's light: inside the hall, so occlusion rays are real work


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.fixture(scope="module")
def atrium(vrt, gpu_device):
    sc = vrt.scene.procedural("atrium", 8, 0, 3)
    assert sc.n_tris == 1048576
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    yield sc, ds
    ds.close()


def _render(vrt, ds, w, h, shadow, light, streams=None, frames=1, y0=0, y1=None):
    """`frames` frames round robin on `streams`; returns the last frame's pixels, hits (occlusion bit split off), colours,
    and all frames' pixel arrays."""
    import torch
    from oracle.pyoracle import HIT_DTYPE
    dev = ds.device
    y1 = h if y1 is None else y1
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = light
    px = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(frames)]
    hits = torch.zeros(h * w * 24, dtype=torch.uint8, device=dev)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    streams = streams or [torch.cuda.current_stream()]
    torch.cuda.synchronize()
    for i in range(frames):
        last = i == frames - 1
        vrt.rtapi.render(ds.accel, w, h, y0, y1, p, px[i].data_ptr(), shadow, hits.data_ptr() if last else None,
                         col.data_ptr() if last else None, cnt.data_ptr() if last else None, streams[i % len(streams)].cuda_stream)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    hn = hits.cpu().numpy().view(HIT_DTYPE).reshape(h, w).copy()
    occ = (hn["blasIdx"] >> 31).astype(bool)
    hn["blasIdx"] &= 0x7FFFFFFF
    return [b.cpu().numpy().view(np.uint32) for b in px], hn, occ, col.cpu().numpy().reshape(h, w, 3), int(cnt.item())


def _occluded_ref(po, sc, w, h, pp, rhits, y0, y1):
    """Occluded pixels of rows [y0,y1) by the FAITHFUL traversal restatement in any-hit mode (same construction of the
    occlusion ray as shadow_ray in csrc/rt_kernels.hip / occluded_toward_light in oracle/rt_oracle.c), from a thread pool."""
    f = np.float32
    rh = rhits[y0:y1].reshape(-1)
    hit_mask = rh["dist"] < 1e29
    rays = po.camera_rays(w, h, y0, y1)[hit_mask]          # (only the rays that hit: a miss's 1e30 would overflow the squares below)
    I = (rays[:, :3] + rays[:, 3:] * rh["dist"][hit_mask].reshape(-1, 1).astype(f)).astype(f)
    L = (np.array(pp.light_pos[:], f)[None] - I).astype(f)
    dist = np.sqrt((L[:, 0] * L[:, 0] + L[:, 1] * L[:, 1]).astype(f) + (L[:, 2] * L[:, 2]).astype(f)).astype(f)
    Ln = (L * (f(1.0) / dist)[:, None]).astype(f)
    srays = np.concatenate([(I + (Ln * f(0.001)).astype(f)).astype(f), Ln], 1).astype(f)
    occ = po.trace_mt(po.trace_faithful, sc, srays, tmax=dist, any_hit=True)
    out = np.zeros(len(rh), bool)
    out[hit_mask] = occ["dist"] < 1e29
    return out.reshape(y1 - y0, w)


_ORACLE_FRAMES = {}


def _spread_rows(h, n_chunks, rows_per_chunk):
    """n_chunks row ranges of rows_per_chunk rows spread evenly over [0, h), the one around the centre row (v == 0: the a-priori
    EXACT launch's pixels) among them."""
    starts = sorted(set(int(round(k * (h - rows_per_chunk) / (n_chunks - 1))) for k in range(n_chunks)) | {h // 2 - rows_per_chunk // 2})
    return [(y, y + rows_per_chunk) for y in starts]


def test_headline_frame_matches_oracle_everywhere(vrt, po, gpu_device, atrium):
    """The frame bench.py times (1,048,576 triangles, 1920x1080, primary + shadow, 4 frames in flight on 4 streams): ALL 2,073,600
    hit records (index, distance bits, barycentrics), the occluded set two-sided against the faithful restatement in any-hit
    mode, pixels and f32 colours of the whole frame."""
    import torch
    sc, ds = atrium
    w, h = 1920, 1080
    vrt.rtapi.accel_frames_in_flight(ds.accel, 4)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(4)]
    try:
        frames, hits, occ, col, nrays = _render(vrt, ds, w, h, 1, LIGHT, streams, frames=8)
    finally:
        vrt.rtapi.accel_frames_in_flight(ds.accel, 1)
    for f in frames[1:]:
        np.testing.assert_array_equal(f, frames[0])
    pp = po.shade_params(light_pos=LIGHT)
    assert nrays == w * h + int((hits["dist"] < 1e29).sum())
    rpx, rhits, rcol, rn = po.render_ex_mt(sc, w, h, pp, 1)
    assert rn == nrays
    assert np.array_equal(_bits(hits), _bits(rhits)), "hit records (index, distance bits, barycentrics)"
    occ_ref = _occluded_ref(po, sc, w, h, pp, rhits, 0, h)
    np.testing.assert_array_equal(occ, occ_ref)
    assert occ_ref.any() and not occ_ref.all()
    np.testing.assert_array_equal(frames[-1], rpx)
    np.testing.assert_allclose(col, rcol, rtol=COLOR_RTOL, atol=0)


def test_the_timed_call_itself_matches_oracle(vrt, po, gpu_device, atrium):
    """What bench.py times: vxrt_render_batch, 1920x1080, 5 frames per set of launches with a light that moves from frame to
    frame, sets alternating on two streams, no optional outputs.  EVERY pixel of frames 0 and 4 of the LAST set against the
    oracle's whole frames (the v == 0 row and the u == 0 column, i.e. the pixels the EXACT launches trace, included); every
    set equal to the first (contexts and streams cannot matter); and the same frames rendered one by one with the hit-record
    output requested give the same pixels."""
    import torch
    sc, ds = atrium
    w, h, n, sets = 1920, 1080, 5, 6
    lights = [(300.0 - 35.0 * f, 480.0 - 12.0 * f, 60.0 + 25.0 * f) for f in range(n)]
    plist = []
    for f in range(n):
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = lights[f]
        plist.append(p)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(2)]
    bufs = [torch.full((n, h, w), 0x5A5A5A5A, dtype=torch.int32, device=gpu_device) for _ in range(sets)]
    cnt = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    vrt.rtapi.accel_frames_in_flight(ds.accel, 2)
    try:
        torch.cuda.synchronize()
        for i in range(sets):
            vrt.rtapi.render_batch(ds.accel, w, h, plist, bufs[i].data_ptr(), h * w, 1, cnt.data_ptr() if i == sets - 1 else None, streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    finally:
        vrt.rtapi.accel_frames_in_flight(ds.accel, 1)
    for b in bufs[1:]:
        assert torch.equal(b, bufs[0])
    assert not torch.equal(bufs[0][0], bufs[0][4])
    last = bufs[-1].cpu().numpy().view(np.uint32)
    rays = 0
    for f in (0, 4):            # every pixel of two frames of the LAST set
        rpx, _, _, rn = po.render_ex_mt(sc, w, h, po.shade_params(light_pos=lights[f]), 1)
        np.testing.assert_array_equal(last[f], rpx)
        rays += rn
    assert rays > 2 * w * h
    # rays counted by the last set's launch: 5 frames, primary + one occlusion ray per hit
    assert 5 * w * h < int(cnt.item()) <= 10 * w * h
    # single frames, with the optional hit-record output: same pixels
    s0 = torch.cuda.current_stream().cuda_stream
    for f in (0, 4):
        one = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
        hits = torch.zeros(h * w * 24, dtype=torch.uint8, device=gpu_device)
        vrt.rtapi.render(ds.accel, w, h, 0, h, plist[f], one.data_ptr(), 1, hits.data_ptr(), None, None, s0)
        torch.cuda.synchronize()
        assert torch.equal(one, bufs[0][f])


def test_a_ranks_batches_with_the_learned_tile_order_match_oracle(vrt, po, gpu_device, atrium):
    """What rank 1 of 4 does in a multi-GPU run of bench.py: batches of 10 frames' shares (interleaved tile rows, 81,600 tiles per
    batch), on two streams.  From a context's second batch on, the tiles are traced longest first in the order learned from the
    batch before (vxrt's LPT for batches of <= 100 K tiles) -- the pixels cannot depend on it: every batch equals the first,
    frame 3's rows equal the same share rendered alone, and ALL of its rows equal the oracle's frame."""
    import torch
    sc, ds = atrium
    w, h, rank, world, n = 1920, 1080, 1, 4, 10
    ig = vrt.sharding.InterleavedGather(h, w, rank, world, gpu_device, slots=1, collective=False, batch=n)
    lights = [(300.0 - 20.0 * f, 480.0 - 8.0 * f, 60.0 + 15.0 * f) for f in range(n)]
    plist = []
    for f in range(n):
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = lights[f]
        plist.append(p)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(2)]
    bufs = [ig.new_frame_buffer(gpu_device) for _ in range(6)]
    vrt.rtapi.accel_frames_in_flight(ds.accel, 2)
    try:
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):      # contexts follow their streams: each context sees three batches, the last two with a learned order
            vrt.rtapi.render_interleaved_batch(ds.accel, w, h, rank, world, plist, b.data_ptr(), ig.frame_stride, 1, None, streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    finally:
        vrt.rtapi.accel_frames_in_flight(ds.accel, 1)
    for b in bufs[1:]:
        assert torch.equal(b, bufs[0])
    rows = vrt.sharding.interleaved_rows(h, rank, world)
    one = torch.zeros((ig.padded_height, w), dtype=torch.int32, device=gpu_device)
    s0 = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_interleaved(ds.accel, w, h, rank, world, plist[3], one.data_ptr(), 1, None, None, None, s0)
    torch.cuda.synchronize()
    assert torch.equal(bufs[-1][3], one)
    # every row of this rank's share of frame 3 against the oracle's frame
    want, _, _, _ = po.render_ex_mt(sc, w, h, po.shade_params(light_pos=lights[3]), 1)
    assert len(rows) >= h // world - 8
    np.testing.assert_array_equal(bufs[-1][3].cpu().numpy().view(np.uint32)[rows], want[rows])


def test_serial_frames_with_learned_tile_order_match_pipelined(vrt, po, gpu_device, atrium):
    """One frame in flight takes the longest-tile-first order from its second frame on; the frame cannot depend on it."""
    sc, ds = atrium
    w, h = 1920, 1080
    frames, hits, occ, col, n = _render(vrt, ds, w, h, 1, LIGHT, frames=3)
    np.testing.assert_array_equal(frames[0], frames[1])
    np.testing.assert_array_equal(frames[0], frames[2])
    pp = po.shade_params(light_pos=LIGHT)
    rpx, rhits, rcol, _ = po.render_ex_mt(sc, w, h, pp, 1)
    np.testing.assert_array_equal(frames[2], rpx)
    assert np.array_equal(_bits(hits), _bits(rhits))


def test_diffuse_bounce_on_the_atrium_matches_oracle(vrt, po, gpu_device, atrium):
    """configs[2] as worded (1 bounce diffuse) at full size: every pixel and f32 colour against orc_render_gi."""
    import torch
    sc, ds = atrium
    w, h = 1920, 1080
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = LIGHT
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    nr = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_diffuse_bounce(ds.accel, w, h, 0, h, p, px.data_ptr(), seed=3, colors_ptr=col.data_ptr(), rays_ptr=nr.data_ptr(), stream=s)
    assert vrt.rtapi.status(s) == 0
    rpx, rcol, rn = po.render_gi_mt(sc, w, h, po.shade_params(light_pos=LIGHT), seed=3)
    assert int(nr.item()) == rn > w * h
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), rcol, rtol=COLOR_RTOL, atol=0)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), rpx)


def test_bunny_class_frame_matches_oracle(vrt, po, gpu_device):
    """configs[1]: ~82k-triangle blob framed to fill the 1024x1024 view, primary + 1 shadow ray."""
    sc = vrt.scene.procedural("bunny", 6, 0, 1)
    assert sc.n_tris == 81920
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w = h = 1024
    light = (20.0, 260.0, -150.0)
    frames, hits, occ, col, nrays = _render(vrt, ds, w, h, 1, light)
    cover = float((hits["dist"] < 1e29).mean())
    assert cover > 0.45, cover                                   # the object fills the view (it was 14 % in round 1)
    assert nrays == w * h + int((hits["dist"] < 1e29).sum())
    pp = po.shade_params(light_pos=light)
    rpx, rhits, rcol, rn = po.render_ex_mt(sc, w, h, pp, 1)
    assert rn == nrays
    assert np.array_equal(_bits(hits), _bits(rhits))
    np.testing.assert_array_equal(occ, _occluded_ref(po, sc, w, h, pp, rhits, 0, h))
    np.testing.assert_array_equal(frames[0], rpx)
    np.testing.assert_allclose(col, rcol, rtol=COLOR_RTOL, atol=0)
    ds.close()


def test_hairball_ao_frame_matches_oracle(vrt, po, gpu_device):
    """configs[4]: 10M-triangle hairball framed to fill the view, 1920x1080, 16 spp ambient occlusion: unoccluded counts,
    colours and pixels of the whole frame against orc_render_ao (same RNG, same IEEE-only sampling recipe)."""
    import torch
    sc = vrt.scene.procedural("hairball_fill", 20000, 250, 7)
    assert sc.n_tris == 10000000
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h, spp = 1920, 1080, 16
    b = sc.bounds
    radius = 0.25 * 0.5 * float(np.linalg.norm(np.array(b[3:]) - np.array(b[:3])))
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (0.0, 400.0, 0.0)
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    cnt = torch.full((h, w), -1, dtype=torch.int32, device=gpu_device)
    nr = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_ao(ds.accel, w, h, 0, h, p, spp, radius, px.data_ptr(), seed=7, colors_ptr=col.data_ptr(),
                        unoccluded_ptr=cnt.data_ptr(), rays_ptr=nr.data_ptr(), stream=s)
    assert vrt.rtapi.status(s) == 0
    rays = int(nr.item())
    hit_px = (rays - w * h) // spp
    assert hit_px > 0.6 * w * h, hit_px / (w * h)                # SURVEY s8d's ray count needs the ball to fill the view
    gcnt, gcol, gpx = cnt.cpu().numpy().view(np.uint32), col.cpu().numpy().reshape(h, w, 3), px.cpu().numpy().view(np.uint32)
    pp = po.shade_params(light_pos=(0.0, 400.0, 0.0))
    rpx, rcol, rcnt, rn = po.render_ao_mt(sc, w, h, pp, spp=spp, radius=radius, seed=7)      # the whole frame: ~25.5 M rays
    assert rn == rays
    np.testing.assert_array_equal(gcnt, rcnt)
    np.testing.assert_allclose(gcol, rcol, rtol=COLOR_RTOL, atol=0)
    np.testing.assert_array_equal(gpx, rpx)
    band = rcnt[h // 2 - 8:h // 2 + 8]
    assert (band < spp).any() and (band > 0).any()
    ds.close()


@pytest.mark.parametrize("world", [2, 8])
def test_4k_frame_as_interleaved_tile_rows_equals_one_shot(vrt, po, gpu_device, atrium, world):
    """configs[3]: 3840x2160 split over `world` ranks the way bench.py --gpus N splits it -- rank r renders the tile rows
    r, r + world, ... (vxrt_render_interleaved) -- here all on one GPU, then assembled with the function the RCCL gather
    path uses.  The assembled frame equals the one-shot frame bit for bit, and the oracle's whole frame, pixel for pixel."""
    import torch
    sc, ds = atrium
    w, h = 3840, 2160
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = LIGHT
    s = torch.cuda.current_stream().cuda_stream
    one = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    cnt = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    vrt.rtapi.render(ds.accel, w, h, 0, h, p, one.data_ptr(), 1, None, None, cnt.data_ptr(), s)
    total = 0
    parts = []
    for r in range(world):
        buf = torch.full((h, w), 0x5A5A5A5A, dtype=torch.int32, device=gpu_device)
        c = torch.zeros(1, dtype=torch.int64, device=gpu_device)
        vrt.rtapi.render_interleaved(ds.accel, w, h, r, world, p, buf.data_ptr(), 1, None, None, c.data_ptr(), s)
        torch.cuda.synchronize()
        total += int(c.item())
        rows = vrt.sharding.interleaved_rows(h, r, world)
        other = np.setdiff1d(np.arange(h), rows)
        assert (buf.cpu().numpy()[other] == 0x5A5A5A5A).all()              # a rank touches only its rows
        parts.append(vrt.sharding.extract_interleaved(buf, h, r, world))
    assert vrt.rtapi.status(s) == 0
    assert total == int(cnt.item())
    frame = vrt.sharding.assemble_interleaved(parts, h, w, world)
    assert torch.equal(frame, one)
    fr = frame.cpu().numpy().view(np.uint32)
    pp = po.shade_params(light_pos=LIGHT)
    if "4k" not in _ORACLE_FRAMES:                 # (the oracle's frame does not depend on the split: traced once for both parametrisations)
        _ORACLE_FRAMES["4k"] = po.render_ex_mt(sc, w, h, pp, 1)[0]
    np.testing.assert_array_equal(fr, _ORACLE_FRAMES["4k"])      # all 8,294,400 pixels


def test_batch_of_frames_equals_the_frames_one_by_one(vrt, po, gpu_device, atrium):
    """vxrt_render_interleaved_batch: 5 frames of one rank's share (rank 1 of 3) with a light that moves from frame to frame, in one
    set of launches -- equal to the five frames rendered one by one, and frame 2 equal to the oracle's on its rows."""
    import torch
    sc, ds = atrium
    w, h, rank, world, n = 488, 270, 1, 3, 5
    ig = vrt.sharding.InterleavedGather(h, w, rank, world, gpu_device, slots=1, collective=False, batch=n)
    plist = []
    for f in range(n):
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = (300.0 - 40.0 * f, 480.0 - 15.0 * f, 60.0 + 30.0 * f)
        plist.append(p)
    s = torch.cuda.current_stream().cuda_stream
    buf = ig.new_frame_buffer(gpu_device)
    buf.fill_(0x5A5A5A5A)
    cnt = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    vrt.rtapi.render_interleaved_batch(ds.accel, w, h, rank, world, plist, buf.data_ptr(), ig.frame_stride, 1, cnt.data_ptr(), s)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(s) == 0
    rows = vrt.sharding.interleaved_rows(h, rank, world)
    other = np.setdiff1d(np.arange(ig.padded_height), rows)
    total = 0
    for f in range(n):
        one = torch.full((ig.padded_height, w), 0x5A5A5A5A, dtype=torch.int32, device=gpu_device)
        c1 = torch.zeros(1, dtype=torch.int64, device=gpu_device)
        vrt.rtapi.render_interleaved(ds.accel, w, h, rank, world, plist[f], one.data_ptr(), 1, None, None, c1.data_ptr(), s)
        torch.cuda.synchronize()
        total += int(c1.item())
        assert torch.equal(buf[f], one)
        assert (buf[f].cpu().numpy()[other] == 0x5A5A5A5A).all()      # only this rank's rows are written
    assert int(cnt.item()) == total
    assert not torch.equal(buf[0], buf[4])                           # the light did move
    opp = po.shade_params(light_pos=(300.0 - 80.0, 480.0 - 30.0, 60.0 + 60.0))
    want, _, _, _ = po.render_ex(sc, w, h, opp, 1)
    assert np.array_equal(buf[2].cpu().numpy().view(np.uint32)[rows], want[rows])


@pytest.mark.parametrize("w,h,rank,world,n", [(173, 99, 0, 2, 3), (64, 40, 4, 5, 2), (200, 120, 0, 1, 7), (96, 8, 0, 3, 32)])
def test_batches_of_ragged_frames(vrt, po, gpu_device, w, h, rank, world, n):
    """Widths and heights that are no multiples of the tile size, more ranks than some frames have tile rows for, one rank
    (whole frames), the largest batch: every frame of the batch equals the same frame rendered alone."""
    import torch
    sc = vrt.scene.procedural("blob", 3, 0, 2)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    vrt.rtapi.accel_frames_in_flight(ds.accel, 2)
    ig = vrt.sharding.InterleavedGather(h, w, rank, world, gpu_device, slots=1, collective=False, batch=n)
    plist = []
    for f in range(n):
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = (100.0 + 13.0 * f, 200.0, -60.0 + 9.0 * f)
        p.background[:] = (0.1 + 0.02 * f, 0.35, 0.25)
        plist.append(p)
    s = torch.cuda.current_stream().cuda_stream
    buf = ig.new_frame_buffer(gpu_device).reshape(n, ig.padded_height, w)
    buf.fill_(0x5A5A5A5A)
    if world == 1:
        vrt.rtapi.render_batch(ds.accel, w, h, plist, buf.data_ptr(), ig.frame_stride, 1, None, s)     # the whole-frame entry point
    else:
        vrt.rtapi.render_interleaved_batch(ds.accel, w, h, rank, world, plist, buf.data_ptr(), ig.frame_stride, 1, None, s)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(s) == 0
    for f in range(n):
        one = torch.full((ig.padded_height, w), 0x5A5A5A5A, dtype=torch.int32, device=gpu_device)
        vrt.rtapi.render_interleaved(ds.accel, w, h, rank, world, plist[f], one.data_ptr(), 1, None, None, None, s)
        torch.cuda.synchronize()
        assert torch.equal(buf[f], one), "frame %d" % f
    if world == 1:
        want, _, _, _ = po.render_ex(sc, w, h, po.shade_params(light_pos=(100.0 + 13.0 * 3, 200.0, -60.0 + 27.0), background=(0.16, 0.35, 0.25)), 1)
        assert np.array_equal(buf[3].cpu().numpy().view(np.uint32)[:h], want)
    ds.close()


def test_twin_full_size_frames_match_oracle(vrt, po, gpu_device):
    """The software twin as tools/config_bench.py times it: the 1,048,576-triangle BVH2 at 1920x1080 through vxrc_render_accel -- wide nodes
    (two BVH2 levels per fetch), frames alternating on two streams (the accel's two frame contexts), every context's tiles traced longest
    first from its second frame on.  Every pixel of every frame, and the colours of the last two, against oracle/rc_oracle.c's whole frame."""
    import torch
    sc = vrt.scene.rc_procedural("atrium", 8, 0, 3)
    assert sc["tri"].size // 36 == 1048576
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    w, h = 1920, 1080
    cam = vrt.scene.rc_camera_like_rtu(w, h)
    light = (300.0, 480.0, 60.0, 1, 1, 1, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25)
    prm = vrt.rtapi.rc_params(cam, light, 1, 1)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(device=gpu_device)]
    frames = 6
    px = [torch.full((h, w), -1, dtype=torch.int32, device=gpu_device) for _ in range(frames)]
    col = [torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device) for _ in range(2)]
    for i in range(frames):
        last = i >= frames - 2
        vrt.rtapi.rc_render_accel(ds.accel, w, h, 0, h, prm, px[i].data_ptr(), col[i % 2].data_ptr() if last else None, streams[i % 2].cuda_stream)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    opx, ocol = po.rc_render_mt(po.rc_args(sc, w, h, cam, light, 1, 1))
    assert (opx != opx[0, 0]).mean() > 0.5
    for i in range(frames):
        np.testing.assert_array_equal(px[i].cpu().numpy().view(np.uint32), opx, err_msg="frame %d" % i)
    for c in col:
        np.testing.assert_allclose(c.cpu().numpy().reshape(h, w, 3), ocol, rtol=COLOR_RTOL, atol=0)
    ds.close()


def test_headline_frame_on_three_shares_behind_one_vx_device(vrt, po, gpu_device, atrium, monkeypatch):
    """VORTEX_HIP_DEVICES (several GPUs behind the one device the reference's host opens) at the headline's size: the 1,048,576-triangle scene
    uploaded through vx_copy_to_dev, 1920x1080 primary + shadow through vx_start on three shares (the one-GPU box repeats device 0: own scene
    copies, layouts, streams, framebuffers), two frames -- every pixel against the oracle's whole frame, MINSTRET = the frame's rays."""
    sc, _ = atrium
    w, h = 1920, 1080
    monkeypatch.setenv("VORTEX_HIP_DEVICES", "0,0,0")
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    monkeypatch.delenv("VORTEX_HIP_DEVICES")
    tr.setup(light_pos=LIGHT, shadow=True)
    assert tr.dev.hip_stat(3) == 3
    a = tr.run()
    rays = tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0)
    b = tr.run()
    assert tr.dev.hip_stat(2) == 2
    tr.close()
    if "headline" not in _ORACLE_FRAMES:
        _ORACLE_FRAMES["headline"] = po.render_ex_mt(sc, w, h, po.shade_params(light_pos=LIGHT), 1)
    want, _, _, want_rays = _ORACLE_FRAMES["headline"]
    np.testing.assert_array_equal(a, want)
    np.testing.assert_array_equal(b, want)
    assert rays == want_rays


def test_the_random_ray_leg_matches_oracle(vrt, po, gpu_device, atrium):
    """north_star's second figure as bench.py runs it: 16 Mi synthetic random rays (torch generator seeded 12345: origins in the scene's box,
    directions on the sphere) against the 1,048,576-triangle BVH through vxrt_trace, ONE launch.  1,048,576 of its hit records -- every 16th
    ray of the buffer, so every part of the launch is sampled -- against the canonical restatement, bit for bit, closest hit; and the first
    262,144 rays in any-hit mode against the faithful one."""
    import torch
    sc, ds = atrium
    n = 16777216
    g = torch.Generator(device=gpu_device).manual_seed(12345)
    lo = torch.tensor(sc.bounds[:3], device=gpu_device)
    hi = torch.tensor(sc.bounds[3:], device=gpu_device)
    o = lo + (hi - lo) * torch.rand((n, 3), generator=g, device=gpu_device)
    d = torch.randn((n, 3), generator=g, device=gpu_device)
    d = d / d.norm(dim=1, keepdim=True)
    rays = torch.cat([o, d], 1).contiguous()
    del o, d
    hits = torch.zeros(n * 24, dtype=torch.uint8, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.trace(ds.accel, rays.data_ptr(), n, hits.data_ptr(), vrt.rtapi.MODE_CLOSEST, None, s)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(s) == 0
    got = hits.view(n, 24)[::16].contiguous().cpu().numpy().view(po.HIT_DTYPE).reshape(-1)
    sample = rays[::16].contiguous().cpu().numpy()
    want = po.trace_mt(po.trace_canonical, sc, sample)
    assert len(want) == 1048576 and 0.3 < (want["dist"] < 1e29).mean() < 1.0      # (hits and misses both)
    assert np.array_equal(_bits(got), _bits(want))
    m = 262144
    vrt.rtapi.trace(ds.accel, rays.data_ptr(), m, hits.data_ptr(), vrt.rtapi.MODE_ANY, None, s)
    torch.cuda.synchronize()
    got_any = hits.view(n, 24)[:m].contiguous().cpu().numpy().view(po.HIT_DTYPE).reshape(-1)
    want_any = po.trace_mt(po.trace_faithful, sc, rays[:m].cpu().numpy(), any_hit=True)
    assert np.array_equal(_bits(got_any), _bits(want_any))


def test_largest_frames(vrt, po, gpu_device):
    """Maximum sizes: a 7680x4320 frame (33 M pixels, 518,400 tiles, an 800 MB hit-record buffer) of the bunny-class scene, primary + shadow --
    72 rows spread over the frame and its last row against the oracle; and the shape check at the limit: 2^25 tiles (2.1 G pixels) are the
    most one set of launches takes -- a window one tile row larger is refused (-1) before anything is allocated or launched."""
    import torch
    sc = vrt.scene.procedural("bunny", 6, 0, 1)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 7680, 4320
    light = (20.0, 260.0, -150.0)
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = light
    px = torch.full((h, w), 0x5A5A5A5A, dtype=torch.int32, device=gpu_device)
    cnt = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render(ds.accel, w, h, 0, h, p, px.data_ptr(), 1, None, None, cnt.data_ptr(), s)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(s) == 0
    got = px.cpu().numpy().view(np.uint32)
    ranges = _spread_rows(h, 8, 8) + [(h - 1, h)]
    want, _, _, _ = po.render_ex_mt(sc, w, h, po.shade_params(light_pos=light), 1, ranges=ranges)
    for y0, y1 in ranges:
        np.testing.assert_array_equal(got[y0:y1], want[y0:y1])
    assert int(cnt.item()) > w * h and (got != 0x5A5A5A5A).all()
    # 8 x 2^25 tiles wide and one tile row high is the limit; nothing is written for a refused window
    one = torch.zeros(16, dtype=torch.int32, device=gpu_device)
    with pytest.raises(RuntimeError):
        vrt.rtapi.render(ds.accel, 8 * (1 << 25) - 7, 9, 0, 9, p, one.data_ptr(), 0, None, None, None, s)
    assert vrt.rtapi.status(s) == 0 and int(one.abs().sum().item()) == 0
    ds.close()
