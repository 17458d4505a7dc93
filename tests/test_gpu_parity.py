"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same inputs.
Bar: hit records bit-exact (distance bits, barycentrics, blasIdx, triIdx); packed RGB8 equal; f32
colour within 1e-5 relative (north_star tolerance; in practice the HIP build is bit-equal)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COLOR_RTOL = 1e-5
FIXTURES = ["teapot", "torus", "sphere", "cone", "cylinder", "cube", "teapot_x3", "sphere_x6", "tex_mix"]


# rays of the sphere_x6 fixture on which the reference expands a TLAS node with a stale base_ptr (rt_traversal.cpp:91-92 after :119):
# excluded from the comparison with the reference's output, compared with the canonical restatement instead
STALE_BASE_RAYS_SPHERE_X6 = 43

def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _hits_np(t):
    from oracle.pyoracle import HIT_DTYPE
    return t.cpu().numpy().view(HIT_DTYPE).reshape(-1)


def gpu_trace(vrt, dscene, rays, mode=0, tmax=None):
    import torch
    n = len(rays)
    r = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).to(dscene.device)
    out = torch.zeros(max(n, 1) * 24, dtype=torch.uint8, device=dscene.device)
    tm = torch.from_numpy(np.ascontiguousarray(tmax, np.float32)).to(dscene.device) if tmax is not None else None
    stream = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.trace(dscene.accel, r.data_ptr() if n else None, n, out.data_ptr() if n else None, mode,
                    tm.data_ptr() if tm is not None else None, stream)
    assert vrt.rtapi.status(stream) == 0
    return _hits_np(out)[:n]


def gpu_render(vrt, dscene, w, h, y0=0, y1=None, shadow=0, params=None):
    import torch
    y1 = h if y1 is None else y1
    dev = dscene.device
    px = torch.full((h, w), 0xDEADBEEF, dtype=torch.int64, device=dev).to(torch.int32)  # sentinel
    hits = torch.zeros(h * w * 24, dtype=torch.uint8, device=dev)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    params = params or vrt.rtapi.default_shade_params()
    vrt.rtapi.render(dscene.accel, w, h, y0, y1, params, px.data_ptr(), shadow, hits.data_ptr(), col.data_ptr(), cnt.data_ptr(), stream)
    assert vrt.rtapi.status(stream) == 0
    hn = _hits_np(hits).reshape(h, w).copy()
    # with shadow != 0 the optional hit output carries the occlusion result in bit 31 of blasIdx (include/vortex_hip.h)
    gpu_render.occluded = (hn["blasIdx"] >> 31).astype(bool)
    hn["blasIdx"] &= 0x7FFFFFFF
    return (px.cpu().numpy().view(np.uint32), hn, col.cpu().numpy().reshape(h, w, 3), int(cnt.item()))


@pytest.mark.parametrize("name", FIXTURES)
def test_trace_matches_reference_fixture(vrt, po, golden, gpu_device, name):
    """Closest hit on the reference-built buffers == what the reference traverser returned."""
    g = golden(name)
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    got = gpu_trace(vrt, ds, g["rays"])
    ok = np.ones(len(got), bool)
    if name == "sphere_x6":   # reference stale-base quirk, see tests/test_oracle_golden.py
        ok = ~po.stale_base_mask(g, g["rays"])
        assert int((~ok).sum()) == STALE_BASE_RAYS_SPHERE_X6      # the mask is a fixed set: a regression cannot hide behind a growing one
        want_fixed, _ = po.trace_canonical(g, g["rays"])
        assert np.array_equal(_bits(got), _bits(want_fixed))
    assert np.array_equal(_bits(got[ok]), _bits(g["hits"][ok]))


@pytest.mark.parametrize("name", FIXTURES)
def test_any_hit_matches_reference_fixture(vrt, po, golden, gpu_device, name):
    g = golden(name)
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    got = gpu_trace(vrt, ds, g["rays"], mode=vrt.rtapi.MODE_ANY)
    ok = ~po.stale_base_mask(g, g["rays"]) if name == "sphere_x6" else np.ones(len(got), bool)
    assert int((~ok).sum()) == (STALE_BASE_RAYS_SPHERE_X6 if name == "sphere_x6" else 0)
    assert np.array_equal(_bits(got[ok]), _bits(g["anyhits"][ok]))


@pytest.mark.parametrize("name", FIXTURES)
def test_shading_matches_reference_fixture(vrt, po, golden, gpu_device, name):
    """Closest-hit / miss shading of the fixture's rays on the GPU == what the reference's own helpers (texSample, diffuseLighting,
    RGB32FtoRGB8, via oracle/_ref) returned: f32 colour within the north-star tolerance (in practice bit-equal), RGB8 equal.
    tex_mix has textured materials with uv below 0 and above 1."""
    import torch
    g = golden(name)
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    n = len(g["rays"])
    r = torch.from_numpy(np.ascontiguousarray(g["rays"], np.float32)).to(gpu_device)
    h = torch.from_numpy(np.ascontiguousarray(g["hits"]).view(np.uint8).copy()).to(gpu_device)   # the reference traverser's hit records
    col = torch.zeros(n * 3, dtype=torch.float32, device=gpu_device)
    px = torch.zeros(n, dtype=torch.int32, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.shade_rays(ds.accel, r.data_ptr(), h.data_ptr(), n, vrt.rtapi.default_shade_params(), col.data_ptr(), px.data_ptr(), s)
    torch.cuda.synchronize()
    np.testing.assert_allclose(col.cpu().numpy().reshape(n, 3), g["colors"], rtol=COLOR_RTOL, atol=0)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), g["rgb8"])
    if name == "tex_mix":
        te = g["triEx"].view(np.float32).reshape(-1, 16)
        assert te[:, 9:15].min() < -1.0 and te[:, 9:15].max() > 2.0
        hit = g["hits"]["dist"] < 1e29
        mats = g["mat"].view(np.int32).reshape(-1, 22)[:, 16]
        tex_hit = mats[te[g["hits"]["triIdx"][hit], 15].view(np.int32)] >= 0
        assert tex_hit.any() and (~tex_hit).any()      # textured and untextured hits both present


@pytest.mark.parametrize("scene_args,w,h", [(("cornell", 0, 0, 1), 256, 256), (("blob", 4, 0, 1), 160, 120),
                                            (("atrium", 5, 0, 3), 200, 112), (("atrium", 5, 0, 3), 67, 45)])
def test_render_matches_oracle(vrt, po, gpu_device, scene_args, w, h):
    """Full frame of the RTU test (ray gen + closest hit + shade + pack), incl. ragged sizes."""
    sc = vrt.scene.procedural(*scene_args)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    px, hits, col, nrays = gpu_render(vrt, ds, w, h)
    rpx, rhits, rcol = po.render(sc, w, h)
    assert nrays == w * h
    assert np.array_equal(_bits(hits), _bits(rhits)), "hit records (index, distance bits, barycentrics)"
    assert np.array_equal(px, rpx), "packed RGB8"
    np.testing.assert_allclose(col, rcol, rtol=COLOR_RTOL, atol=0)
    assert (rhits["dist"] < 1e29).any() and (rhits["dist"] >= 1e29).any() or scene_args[0] != "blob"


def test_render_row_window_only_touches_its_rows(vrt, po, gpu_device):
    sc = vrt.scene.procedural("blob", 3, 0, 1)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 96, 80
    rpx, _, _ = po.render(sc, w, h)
    for (y0, y1) in ((0, 40), (40, 80), (24, 27), (79, 80), (10, 10)):
        px, _, _, n = gpu_render(vrt, ds, w, h, y0, y1)
        assert n == (y1 - y0) * w
        assert np.array_equal(px[y0:y1], rpx[y0:y1])
        assert (px[:y0] == 0xDEADBEEF).all() and (px[y1:] == 0xDEADBEEF).all()


def test_random_rays_on_procedural_scene(vrt, po, gpu_device):
    """Incoherent rays (the headline workload shape) against our own builder's tree."""
    sc = vrt.scene.procedural("atrium", 5, 0, 7)
    lo, hi = sc.bounds[:3], sc.bounds[3:]
    rng = np.random.default_rng(12345)
    n = 20000
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    got = gpu_trace(vrt, ds, rays)
    want, st = po.trace_faithful(sc, rays)
    assert st["trail_overflow"] == 0
    assert np.array_equal(_bits(got), _bits(want))
    got_any = gpu_trace(vrt, ds, rays, mode=vrt.rtapi.MODE_ANY)
    want_any, _ = po.trace_faithful(sc, rays, any_hit=True)
    assert np.array_equal(_bits(got_any), _bits(want_any))


def test_degenerate_rays_and_tmax(vrt, po, golden, gpu_device):
    """Zero direction components (inf reciprocals, NaN slabs), zero-length and NaN rays, tmax cut."""
    g = golden("teapot")
    rays = g["rays"][:512].copy()
    rays[0:64, 3] = 0.0
    rays[64:128, 4] = 0.0
    rays[128:160, 3:5] = 0.0
    rays[160:164, 3:6] = 0.0
    rays[164:168, 0] = np.nan
    rays[168:172, 5] = np.inf
    tmax = np.full(len(rays), 1e30, np.float32)
    tmax[256:] = g["hits"]["dist"][256:512] * np.float32(0.75)
    tmax[256:][g["hits"]["dist"][256:512] >= 1e29] = 3.0
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    got = gpu_trace(vrt, ds, rays, tmax=tmax)
    want, _ = po.trace_faithful(g, rays, tmax=tmax)
    assert np.array_equal(_bits(got), _bits(want))


def test_empty_and_bad_arguments(vrt, golden, gpu_device):
    g = golden("sphere")
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    assert len(gpu_trace(vrt, ds, np.zeros((0, 6), np.float32))) == 0
    bad = vrt.rtapi.VxrtScene()
    C.memmove(C.byref(bad), C.byref(ds.c), C.sizeof(bad))
    bad.n_tris = 0
    with pytest.raises(vrt.runtime.VxError):
        vrt.rtapi.accel_build(bad)
    with pytest.raises(vrt.runtime.VxError):
        vrt.rtapi.trace(ds.accel, None, 4, None, 0, None, None)   # n > 0 without buffers
    with pytest.raises(vrt.runtime.VxError):
        vrt.rtapi.trace(ds.accel, 1, 4, 1, 7, None, None)         # unknown mode
    with pytest.raises(vrt.runtime.VxError):
        vrt.rtapi.trace(None, 1, 4, 1, 0, None, None)             # no accel


def test_malformed_trees_are_rejected_at_build_time(vrt, golden, gpu_device):
    """Every index the traversal can follow is validated when the acceleration layout is built, so a
    corrupt scene fails on the host (-1) instead of faulting the GPU."""
    import torch
    g = golden("teapot")
    node = np.dtype([("o", "<f4", 3), ("e", "i1", 3), ("imask", "u1"), ("lf", "<u4"), ("ld", "<u4"), ("ch", "u1", (4, 7))])

    def try_build(mut):
        sc = {k: v.copy() for k, v in g.items() if k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
        mut(sc)
        return vrt.tracer.DeviceScene(sc, gpu_device)

    try_build(lambda sc: None).close()
    nodes = g["bvh"].view(node)
    internal = int(np.nonzero(nodes["ld"] == 0)[0][0])
    leaf = int(np.nonzero(nodes["ld"] != 0)[0][0])

    def child_out_of_range(sc):
        sc["bvh"].view(node)["lf"][internal] = 0x7FFFFFF0

    def child_before_parent(sc):   # would be a cycle
        n = sc["bvh"].view(node)
        i2 = int(np.nonzero((n["ld"] == 0) & (np.arange(len(n)) > 4))[0][0])
        n["lf"][i2] = 0

    def leaf_past_triangles(sc):
        sc["bvh"].view(node)["lf"][leaf] = len(sc["tri"]) // 36

    def wrong_kind(sc):
        sc["bvh"].view(node)["imask"][internal] = 1

    def bad_instance(sc):
        sc["tlas"].view(node)["ld"][0] = 5

    def bad_bvh_offset(sc):
        sc["blas"].view(np.uint32)[0] = 0x7FFFFFFF

    for mut in (child_out_of_range, child_before_parent, leaf_past_triangles, wrong_kind, bad_instance, bad_bvh_offset):
        with pytest.raises(vrt.runtime.VxError):
            try_build(mut)
    torch.cuda.synchronize()


def test_malformed_shading_inputs_are_rejected_at_build_time(vrt, golden, gpu_device):
    """closest.cpp:52-77 dereferences mat[texId] and the texels of a textured material unchecked; the accel build validates
    them, so a scene whose shading would read outside its buffers fails on the host (-1) instead of faulting the GPU."""
    import torch
    g = golden("tex_mix")
    mat_dt = np.dtype([("f", "<f4", 16), ("tex_id", "<i4"), ("illum", "<i4"), ("tw", "<u4"), ("th", "<u4"), ("off", "<u8")])
    assert mat_dt.itemsize == 88

    def try_build(mut, drop_tex=False):
        sc = {k: v.copy() for k, v in g.items() if k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
        mut(sc)
        if drop_tex:
            sc["tex"] = np.zeros(0, np.uint8)
        return vrt.tracer.DeviceScene(sc, gpu_device)

    try_build(lambda sc: None).close()
    n_mats = len(g["mat"]) // 88
    textured = int(np.nonzero(g["mat"].view(mat_dt)["tex_id"] >= 0)[0][0])

    def tex_id_out_of_range(sc):
        sc["triEx"].view(np.uint32).reshape(-1, 16)[7, 15] = n_mats

    def zero_width(sc):
        sc["mat"].view(mat_dt)["tw"][textured] = 0

    def zero_height(sc):
        sc["mat"].view(mat_dt)["th"][textured] = 0

    def offset_past_buffer(sc):
        sc["mat"].view(mat_dt)["off"][textured] = len(sc["tex"]) - 16

    def huge_dimensions(sc):
        sc["mat"].view(mat_dt)["tw"][textured] = 0x10000
        sc["mat"].view(mat_dt)["th"][textured] = 0x10000

    for mut in (tex_id_out_of_range, zero_width, zero_height, offset_past_buffer, huge_dimensions):
        with pytest.raises(vrt.runtime.VxError):
            try_build(mut)
    with pytest.raises(vrt.runtime.VxError):          # textured material, no texture buffer at all
        try_build(lambda sc: None, drop_tex=True)
    torch.cuda.synchronize()


def _occlusion_oracle(po, sc, w, h, pp, rhits):
    """Occluded set of a frame decided by the faithful restatement in any-hit mode: occlusion rays built exactly as the
    kernel documents them (origin I + 0.001 L, direction L, tmax |light - I|; shadow_ray in csrc/rt_kernels.hip)."""
    f = np.float32
    hit_mask = rhits["dist"].reshape(-1) < 1e29
    rays = po.camera_rays(w, h)[hit_mask]                   # (only the rays that hit: a miss's 1e30 would overflow the squares below)
    I = (rays[:, :3] + rays[:, 3:] * rhits["dist"].reshape(-1)[hit_mask].reshape(-1, 1).astype(f)).astype(f)
    L = (np.array(pp.light_pos[:], f)[None] - I).astype(f)
    dist = np.sqrt((L[:, 0] * L[:, 0] + L[:, 1] * L[:, 1]).astype(f) + (L[:, 2] * L[:, 2]).astype(f)).astype(f)
    Ln = (L * (f(1.0) / dist)[:, None]).astype(f)
    srays = np.concatenate([(I + (Ln * f(0.001)).astype(f)).astype(f), Ln], 1).astype(f)
    occ, _ = po.trace_faithful(sc, srays, tmax=dist, any_hit=True)
    out = np.zeros(w * h, bool)
    out[hit_mask] = occ["dist"] < 1e29
    return out.reshape(h, w), hit_mask.reshape(h, w)


@pytest.mark.parametrize("scene_args,w,h,light", [(("blob", 4, 0, 1), 128, 96, (60.0, 260.0, -150.0)),
                                                  (("atrium", 5, 0, 3), 200, 112, (300.0, 480.0, 60.0))])
def test_shadow_rays_extension(vrt, po, gpu_device, scene_args, w, h, light):
    """primary + 1 shadow ray (the headline workload's shape; BASELINE configs 2/3; no reference counterpart).  Two-sided:
    the whole frame -- pixels, colours, hit records, ray total -- equals the restatement's frame with the shadow extension,
    and the set of occluded pixels the kernel reports equals what the faithful traversal restatement decides in any-hit
    mode.  The timed kernel takes the UNORDERED any-hit path for these rays; which triangle is met first cannot change
    the boolean, and this test is what pins that."""
    sc = vrt.scene.procedural(*scene_args)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = light
    px, hits, col, nrays = gpu_render(vrt, ds, w, h, shadow=1, params=p)
    occ_gpu = gpu_render.occluded
    pp = po.shade_params(light_pos=tuple(p.light_pos))
    rpx, rhits, rcol, rn = po.render_ex(sc, w, h, pp, 1)
    assert np.array_equal(_bits(hits), _bits(rhits))
    assert nrays == rn == w * h + int((rhits["dist"] < 1e29).sum())
    occ_ref, hit_mask = _occlusion_oracle(po, sc, w, h, pp, rhits)
    assert occ_ref.any() and (hit_mask & ~occ_ref).any()
    np.testing.assert_array_equal(occ_gpu, occ_ref)
    np.testing.assert_array_equal(px, rpx)
    np.testing.assert_allclose(col, rcol, rtol=COLOR_RTOL, atol=0)
    # and the shadow really changes the frame: an occluded pixel with N.L > 0 is darker than the unshadowed restatement
    _, _, ucol = po.render(sc, w, h, pp)
    assert (col[occ_ref] <= ucol[occ_ref] + 1e-7).all() and (col[occ_ref] < ucol[occ_ref] - 1e-4).any()
    np.testing.assert_allclose(col[~occ_ref], ucol[~occ_ref], rtol=COLOR_RTOL, atol=0)


def test_fetch_counters_equal_oracle_counts(vrt, po, gpu_device):
    """The counting build of the render kernel reports N_node / N_inst / N_tri per launch (inputs of
    roofline.achieved): they must equal what the canonical CPU restatement counts on the same frame."""
    import torch
    sc = vrt.scene.procedural("atrium", 5, 0, 3)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 160, 96
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    c = vrt.rtapi.render_stats(ds.accel, w, h, 0, h, vrt.rtapi.default_shade_params(), px.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    rays = po.camera_rays(w, h)
    hits, st = po.trace_canonical(sc, rays)
    assert c["rays"] == w * h and c["pixels"] == w * h
    assert c["node_fetches"] == st["node_reads"]
    assert c["inst_fetches"] == st["inst_reads"]
    assert c["tri_fetches"] == st["tri_reads"]
    assert c["shaded_hits"] == int((hits["dist"] < 1e29).sum())
    rpx, _, _ = po.render(sc, w, h)
    assert np.array_equal(px.cpu().numpy().view(np.uint32), rpx)


def test_frames_in_flight_do_not_change_results(vrt, po, gpu_device):
    """vxrt_accel_frames_in_flight: frames issued round robin on three streams (with shadow rays, so the
    hit-record buffer and the deferred-ray list of a frame are live across kernels) equal the serial
    frames; a context reused on another stream is ordered behind its previous frame."""
    import torch
    sc = vrt.scene.procedural("blob", 4, 0, 5)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 320, 200
    lights = [(40.0, 60.0, 20.0), (-30.0, 50.0, 10.0), (5.0, 80.0, -40.0), (60.0, 20.0, 60.0), (0.0, 100.0, 0.0)]
    plist = []
    for lp in lights:
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = lp
        plist.append(p)
    serial = [gpu_render(vrt, ds, w, h, shadow=1, params=p)[0] for p in plist]
    assert any((serial[0] != s).any() for s in serial[1:])      # the frames really differ
    with pytest.raises(RuntimeError):
        vrt.rtapi.accel_frames_in_flight(ds.accel, 0)
    with pytest.raises(RuntimeError):
        vrt.rtapi.accel_frames_in_flight(ds.accel, 9)
    vrt.rtapi.accel_frames_in_flight(ds.accel, 3)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(2)]   # fewer streams than contexts on purpose
    outs = [torch.zeros((h, w), dtype=torch.int32, device=gpu_device) for _ in range(2 * len(plist))]
    torch.cuda.synchronize()
    for i, buf in enumerate(outs):
        vrt.rtapi.render(ds.accel, w, h, 0, h, plist[i % len(plist)], buf.data_ptr(), 1, None, None, None,
                         streams[i % len(streams)].cuda_stream)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    for i, buf in enumerate(outs):
        np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), serial[i % len(plist)])
    vrt.rtapi.accel_frames_in_flight(ds.accel, 1)
    again = gpu_render(vrt, ds, w, h, shadow=1, params=plist[0])[0]
    np.testing.assert_array_equal(again, serial[0])


def test_one_context_moving_between_streams_keeps_the_frames_ordered(vrt, po, gpu_device):
    """An accel with one frame context that has only seen one stream records no completion event per frame; its first frame on ANOTHER
    stream must still wait for the frames before it (the context's hit buffer and control block are shared), and from then on the
    event orders them: frames alternating between three streams, never synchronised in between, equal the serial ones."""
    import torch
    sc = vrt.scene.procedural("blob", 4, 0, 1)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 320, 200
    plist = []
    for lp in ((40.0, 60.0, 20.0), (-30.0, 50.0, 10.0), (5.0, 80.0, -40.0)):
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = lp
        plist.append(p)
    serial = [gpu_render(vrt, ds, w, h, shadow=1, params=p)[0] for p in plist]      # (the default stream: the accel's first)
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(3)]
    outs = [torch.zeros((h, w), dtype=torch.int32, device=gpu_device) for _ in range(24)]
    torch.cuda.synchronize()
    for i, buf in enumerate(outs):
        vrt.rtapi.render(ds.accel, w, h, 0, h, plist[i % 3], buf.data_ptr(), 1, None, None, None, streams[(i * 5 + i // 4) % 3].cuda_stream)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(streams[0].cuda_stream) == 0
    for i, buf in enumerate(outs):
        np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), serial[i % 3])
    ds.close()


def test_inverted_child_boxes_take_the_generic_slab_form(vrt, po, golden, gpu_device):
    """The fast slab test picks the near/far plane by the sign of 1/d, which presumes q_lo <= q_hi; a
    tree with inverted child boxes (legal bytes for the reference, which just evaluates min/max) must
    make the accel build select the generic form, and the hits must still equal the reference
    algorithm's on that tree."""
    g = golden("torus")
    node = np.dtype([("o", "<f4", 3), ("e", "i1", 3), ("imask", "u1"), ("lf", "<u4"), ("ld", "<u4"), ("ch", "u1", (4, 7))])
    sc = {k: v.copy() for k, v in g.items() if k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    n = sc["bvh"].view(node)
    internal = np.nonzero(n["ld"] == 0)[0]
    swapped = 0
    for i in internal[1::3]:          # swap lo.x <-> hi.x of one child in a third of the internal nodes
        ch = n["ch"][i]
        valid = np.nonzero(ch[:, 0] != 0)[0]
        if len(valid) == 0:               # unused slot of the reference-built buffer
            continue
        k = int(valid[-1])
        if ch[k, 1] != ch[k, 4]:
            ch[k, 1], ch[k, 4] = ch[k, 4], ch[k, 1]
            swapped += 1
    assert swapped > 10
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    got = gpu_trace(vrt, ds, g["rays"])
    want, _ = po.trace_faithful(sc, g["rays"])
    np.testing.assert_array_equal(_bits(got), _bits(want))
    assert (want["dist"] < 1e29).sum() > 0
    # min/max are symmetric, so the reference finds the same hits as on the original tree; a sign-selected
    # test applied to these boxes would take far planes for near ones and lose most of them
    base, _ = po.trace_faithful(g, g["rays"])
    np.testing.assert_array_equal(_bits(base), _bits(want))


@pytest.mark.parametrize("depth,shadow", [(1, 0), (2, 0), (4, 0), (4, 1), (16, 0)])
def test_mirror_bounce_matches_oracle(vrt, po, gpu_device, depth, shadow):
    """closest.cpp:95-121: reflective instances spawn a mirror ray while bounce + 1 < max_depth.  The
    kernels run it as a wavefront over depth levels and fold the colours back in the reference's order
    of operations: pixels and colours equal the recursive restatement (which is unpinned for this arm:
    the reference's shaders only build for RISC-V and its scenes have reflectivity 0)."""
    from scenes import mirror_hall
    b = mirror_hall(vrt)
    ds = vrt.tracer.DeviceScene(b, gpu_device)
    w, h = 160, 96
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (150.0, 220.0, -60.0)
    p.max_depth = depth
    px, hits, col, nrays = gpu_render(vrt, ds, w, h, shadow=shadow, params=p)
    pp = po.shade_params(light_pos=tuple(p.light_pos), max_depth=depth)
    rpx, rhits, rcol, rn = po.render_ex(b, w, h, pp, shadow)
    assert np.array_equal(_bits(hits), _bits(rhits))
    assert nrays == rn
    np.testing.assert_allclose(col, rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(px, rpx)
    if depth > 1:
        flat, _, fcol, fn = po.render_ex(b, w, h, po.shade_params(light_pos=tuple(p.light_pos), max_depth=1), shadow)
        assert rn > fn and (rcol != fcol).any()          # the bounce really contributes


@pytest.mark.parametrize("name", ["mirror_teapot", "mirror_trio"])
def test_mirror_bounce_matches_the_reference_twin_fixture(vrt, po, golden, gpu_device, name):
    """The HIP mirror-bounce path against REFERENCE OBJECT CODE: tests/golden/mirror_*.npz holds what the reference's software twin
    (raycast/render.h:210-277) returns for the RTU camera rays on a scene both reference builders built (reflective instances 0.5 /
    0.3, oracle/gen_golden_mirror.py).  vxrt_render with max_depth 1..4 on the reference-built RTU buffers: f32 colours within 1e-5
    relative of the twin's, RGB8 equal in every pixel (no truncation flips on these frames: asserted, not assumed)."""
    g = golden(name)
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    w, h = int(g["width"]), int(g["height"])
    L = g["light12"]
    for d in [int(x) for x in g["depths"]]:
        p = vrt.rtapi.default_shade_params()
        p.ambient[:] = tuple(L[6:9]); p.light_color[:] = tuple(L[3:6]); p.light_pos[:] = tuple(L[0:3]); p.background[:] = tuple(L[9:12])
        p.max_depth = d
        px, hits, col, nrays = gpu_render(vrt, ds, w, h, shadow=0, params=p)
        np.testing.assert_allclose(col, g["colors_d%d" % d], rtol=COLOR_RTOL, atol=0)
        assert np.array_equal(px, g["rgb8_d%d" % d]), "%d RGB8 pixels differ from the reference twin at depth %d" % (int((px != g["rgb8_d%d" % d]).sum()), d)
    assert not np.array_equal(g["rgb8_d1"], g["rgb8_d2"])
    ds.close()


@pytest.mark.parametrize("spp,radius", [(1, 25.0), (4, 60.0), (16, 15.0)])
def test_ambient_occlusion_pass_matches_oracle(vrt, po, gpu_device, spp, radius):
    """vxrt_render_ao (extension for BASELINE config 5): the sampling recipe uses the reference RNG
    (common.h:129-147) and IEEE add/mul/div/sqrt only, so device and checker generate the same occlusion
    rays: unoccluded counts, ray totals and pixels are equal, colours within the colour tolerance."""
    import torch
    from scenes import mirror_hall
    b = mirror_hall(vrt, 0.0, 0.0)
    ds = vrt.tracer.DeviceScene(b, gpu_device)
    w, h = 144, 88
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (150.0, 220.0, -60.0)
    dev = gpu_device
    px = torch.zeros((h, w), dtype=torch.int32, device=dev)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    cnt = torch.full((h, w), -1, dtype=torch.int32, device=dev)
    nr = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_ao(ds.accel, w, h, 0, h, p, spp, radius, px.data_ptr(), seed=7, colors_ptr=col.data_ptr(),
                        unoccluded_ptr=cnt.data_ptr(), rays_ptr=nr.data_ptr(), stream=stream)
    assert vrt.rtapi.status(stream) == 0
    rpx, rcol, rcnt, rn = po.render_ao(b, w, h, po.shade_params(light_pos=tuple(p.light_pos)), spp=spp, radius=radius, seed=7)
    np.testing.assert_array_equal(cnt.cpu().numpy().view(np.uint32), rcnt)
    assert int(nr.item()) == rn
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), rpx)
    hit = rn > w * h
    assert hit and (rcnt < spp).any() and (rcnt == spp).any()     # some occlusion, some open sky
    with pytest.raises(vrt.runtime.VxError):
        vrt.rtapi.render_ao(ds.accel, w, h, 0, h, p, 0, radius, px.data_ptr(), stream=stream)


def test_trace_fetch_counters_equal_oracle_counts(vrt, po, gpu_device):
    """vxrt_trace_stats: the counts behind the random-ray leg's bytes per ray equal the canonical restatement's."""
    import torch
    sc = vrt.scene.procedural("atrium", 5, 0, 3)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    rng = np.random.default_rng(77)
    b = np.array(sc.bounds, np.float32)
    n = 20000
    o = b[:3] + (b[3:] - b[:3]) * rng.random((n, 3), dtype=np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    r = torch.from_numpy(rays).to(gpu_device)
    out = torch.zeros(n * 24, dtype=torch.uint8, device=gpu_device)
    c = vrt.rtapi.trace_stats(ds.accel, r.data_ptr(), n, out.data_ptr(), vrt.rtapi.MODE_CLOSEST, None, torch.cuda.current_stream().cuda_stream)
    hits, st = po.trace_canonical(sc, rays)
    assert c["rays"] == n
    assert c["node_fetches"] == st["node_reads"] and c["inst_fetches"] == st["inst_reads"] and c["tri_fetches"] == st["tri_reads"]
    assert c["bytes"] == 48 * n + 52 * (st["node_reads"] + st["inst_reads"]) + 36 * st["tri_reads"]
    assert np.array_equal(_bits(_hits_np(out)[:n]), _bits(hits))


def test_learned_tile_order_does_not_change_results(vrt, po, gpu_device):
    """From the second frame of a window on, a context with one frame in flight starts its most expensive tiles
    first (cost learned from the frame before; frames of >= 20000 tiles only).  The order of tiles cannot change a
    pixel: frames 1..4 of a 1920x1080 window are identical, and a band of rows equals the oracle."""
    sc = vrt.scene.procedural("atrium", 5, 0, 3)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 1916, 1076        # partial tiles on both edges, 32400 tiles
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (300.0, 480.0, 60.0)
    frames = [gpu_render(vrt, ds, w, h, shadow=1, params=p) for _ in range(4)]
    for px, hits, col, n in frames[1:]:
        np.testing.assert_array_equal(px, frames[0][0])
        assert np.array_equal(_bits(hits), _bits(frames[0][1])) and n == frames[0][3]
    y0, y1 = 500, 508
    rpx, rhits, _ = po.render(sc, w, h, po.shade_params(light_pos=tuple(p.light_pos)), y0=y0, y1=y1)
    assert np.array_equal(_bits(frames[3][1][y0:y1]), _bits(rhits.reshape(h, w)[y0:y1]))
    band = [gpu_render(vrt, ds, w, h, y0=40, y1=1000, shadow=1, params=p)[0] for _ in range(3)]   # another window: order relearned
    for b in band:
        np.testing.assert_array_equal(b[40:1000], frames[0][0][40:1000])


def test_diffuse_bounce_pass_matches_oracle(vrt, po, gpu_device):
    """vxrt_render_diffuse_bounce (extension for BASELINE config 3's "1 bounce diffuse"): same trig-free sampling
    recipe as the AO pass, one closest-hit bounce ray per primary hit; pixels, colours and ray totals equal the checker's."""
    import torch
    from scenes import mirror_hall
    b = mirror_hall(vrt, 0.0, 0.0)
    ds = vrt.tracer.DeviceScene(b, gpu_device)
    w, h = 152, 96
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (150.0, 220.0, -60.0)
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    nr = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_diffuse_bounce(ds.accel, w, h, 0, h, p, px.data_ptr(), seed=11, colors_ptr=col.data_ptr(), rays_ptr=nr.data_ptr(), stream=s)
    assert vrt.rtapi.status(s) == 0
    pp = po.shade_params(light_pos=tuple(p.light_pos))
    rpx, rcol, rn = po.render_gi(b, w, h, pp, seed=11)
    assert int(nr.item()) == rn > w * h
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), rpx)
    _, _, direct = po.render(b, w, h, pp)
    assert (rcol >= direct - 1e-7).all() and (rcol > direct + 1e-3).any()      # the bounce only adds light


@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 2049, 20000])
def test_exact_launch_with_nothing_deferred_takes_no_queue_position(vrt, po, golden, gpu_device, n):
    """Guard of round 2's abort (a fault in the EXACT launch of a ray buffer with nothing deferred): with every ray inside the fast
    domain the main launch defers nothing, and the EXACT launch over the (empty) deferred list must find every queue shard past
    the end of its job range WITHOUT touching the shard's counter -- its eight counters stay zero, whatever the buffer size
    (fewer rays than one wavefront included).  The main launch's counters show the test looks at the right block."""
    g = golden("teapot")
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    rs = np.random.RandomState(n)
    rays = g["rays"][rs.randint(0, len(g["rays"]), size=n)].astype(np.float32).copy()
    rays[:, 3:] += rs.uniform(1e-4, 2e-4, size=(n, 3)).astype(np.float32)      # no zero direction component: nothing for the EXACT launch
    inv = 1.0 / rays[:, 3:]
    assert np.isfinite(inv).all() and (np.abs(inv) <= 2.0 ** 64).all()
    got = gpu_trace(vrt, ds, rays)
    want, _ = po.trace_canonical(g, rays)
    assert np.array_equal(_bits(got), _bits(want))
    import torch
    ctl = vrt.rtapi.debug_read_control(ds.accel, 0, 800, torch.cuda.current_stream().cuda_stream)
    main_q = ctl[32:32 + 256:32]
    exact_q = ctl[32 + 256:32 + 512:32]
    assert ctl[0] == 0, "rays deferred to the EXACT launch: %d" % ctl[0]
    assert int(main_q.sum()) >= n, main_q                  # the main launch did draw its jobs from these counters
    assert not exact_q.any(), "EXACT launch touched queue counters with nothing to do: %s" % exact_q
    ds.close()


def test_exact_launch_takes_exactly_the_deferred_rays(vrt, po, golden, gpu_device):
    """The other side of the guard: axis-parallel rays (zero direction components) are ALL deferred; the EXACT launch's counters then
    cover the deferred list and nothing of the shards past its end."""
    g = golden("teapot")
    ds = vrt.tracer.DeviceScene(g, gpu_device)
    n = 300
    rays = g["rays"][:n].astype(np.float32).copy()
    rays[:, 3:] = 0.0
    rays[np.arange(n), 3 + (np.arange(n) % 3)] = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    got = gpu_trace(vrt, ds, rays)
    want, _ = po.trace_canonical(g, rays)
    assert np.array_equal(_bits(got), _bits(want))
    import torch
    ctl = vrt.rtapi.debug_read_control(ds.accel, 0, 800, torch.cuda.current_stream().cuda_stream)
    assert ctl[0] == n
    exact_q = ctl[32 + 256:32 + 512:32].astype(np.int64)
    per_shard = ((n + 7) // 8 + 63) & ~63          # the kernel's split of the deferred list over 8 shards
    used = -(-n // per_shard)
    assert (exact_q[:used] >= np.minimum(per_shard, n - per_shard * np.arange(used))).all() and not exact_q[used:].any(), exact_q
    ds.close()


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_quirks_mode_equals_the_reference_on_every_ray(vrt, po, golden, gpu_device, name):
    """vxrt_trace_reference_quirks: the RTU's traversal restated literally on a flat memory image -- trail, short stack, restart,
    re-descent, and the stale base_ptr of rt_traversal.cpp:91-92.  EVERY ray of every fixture equals what the reference's own
    traverser returned (closest hit and first accepted candidate), UNMASKED: including the 43 rays of sphere_x6 on which the
    reference reads unrelated nodes and the canonical kernels (the default) deliberately differ."""
    import torch
    g = golden(name)
    img = po.Image(g)
    mem = torch.from_numpy(img.mem).to(gpu_device)
    rays = torch.from_numpy(np.ascontiguousarray(g["rays"], np.float32)).to(gpu_device)
    n = len(g["rays"])
    offs = (img.off["tlas"], img.off["blas"], img.off["bvh"], img.off["tri"])
    s = torch.cuda.current_stream().cuda_stream
    for mode, key in ((vrt.rtapi.MODE_CLOSEST, "hits"), (vrt.rtapi.MODE_ANY, "anyhits")):
        out = torch.zeros(n * 24, dtype=torch.uint8, device=gpu_device)
        vrt.rtapi.trace_reference_quirks(mem.data_ptr(), mem.numel(), offs, rays.data_ptr(), n, out.data_ptr(), mode, None, s)
        assert vrt.rtapi.status(s) == 0
        got = _hits_np(out)[:n]
        assert np.array_equal(_bits(got), _bits(g[key])), "%s / %s: %d rays differ" % (name, key, int((_bits(got).reshape(n, -1) != _bits(g[key]).reshape(n, -1)).any(1).sum()))
    if name == "sphere_x6":
        # ... and there the default (canonical) kernels do differ, on exactly the masked rays
        ds = vrt.tracer.DeviceScene(g, gpu_device)
        canon = gpu_trace(vrt, ds, g["rays"])
        differ = (_bits(canon).reshape(n, -1) != _bits(g["hits"]).reshape(n, -1)).any(1)
        mask = po.stale_base_mask(g, g["rays"])
        assert differ.any() and not (differ & ~mask).any()
        ds.close()


@pytest.mark.gpu
def test_identity_instance_start_and_signed_zero_origins(vrt, po, gpu_device):
    """A single instance whose inverse transform is the identity lets a ray start at the BLAS root with the world ray as it is (start_ray):
    the reference's matrix arithmetic would return the same bits -- except for an origin component that is -0, which it turns into +0; such
    rays take the general instance step.  Rays through vertices and along axis planes of a box at the origin, with every +0 / -0 / non-zero
    pattern of the origin, against the faithful restatement; then the same triangles under a TRANSLATED instance (no identity: the general
    step for every ray) and under an identity whose zeros are -0."""
    sc = vrt.scene.procedural("cornell")
    rng = np.random.default_rng(5)
    lo, hi = np.array(sc.bounds[:3]), np.array(sc.bounds[3:])
    n = 4096
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pat = rng.integers(0, 3, size=(n, 3))                      # per component: keep / +0 / -0
    o[pat == 1] = 0.0
    o[pat == 2] = -0.0
    assert (np.signbit(o) & (o == 0)).any() and ((o == 0) & ~np.signbit(o)).any()
    rays = np.concatenate([o, d], 1).astype(np.float32)
    for variant in ("identity", "translated", "negative zeros"):
        bufs = dict(sc.buffers)
        blas = sc["blas"].copy().view(np.float32)
        if variant == "translated":
            blas[1 + 3] -= 7.0        # invTransform[0][3]
            blas[17 + 3] += 7.0       # transform[0][3]
        elif variant == "negative zeros":
            m = blas[1:13]
            m[m == 0] = -0.0
        bufs["blas"] = blas.view(np.uint8)
        s2 = vrt.scene.Scene(bufs)
        ds = vrt.tracer.DeviceScene(s2, gpu_device)
        for mode, any_hit in ((vrt.rtapi.MODE_CLOSEST, False), (vrt.rtapi.MODE_ANY, True)):
            got = gpu_trace(vrt, ds, rays, mode=mode)
            want, _ = po.trace_faithful(s2, rays, any_hit=any_hit)
            assert np.array_equal(_bits(got), _bits(want)), (variant, mode)
        assert (want["dist"] < 1e29).any()
        ds.close()


def test_wave_log_reports_every_slot_it_documents(vrt, po, gpu_device):
    """vxrt_render_wave_log (diagnostic build, include/vortex_hip.h): 16 u64 per wavefront.  The tools that explain a launch's tail
    (tools/wave_balance.py, wave_balance_batch.py) divide by slots 10-12 -- shader clocks inside the node body, the instance + leaf
    part, and the whole wavefront -- so those must be written and consistent; the frame the build renders is still the oracle's."""
    import ctypes as C
    import torch
    sc = vrt.scene.procedural("atrium", 5, 0, 3)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 320, 200
    L = vrt.rtapi._lib()
    L.vxrt_render_wave_log.restype = C.c_int
    L.vxrt_render_wave_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vrt.rtapi.ShadeParams), C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    p = vrt.rtapi.default_shade_params()
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    cnt = torch.zeros(8, dtype=torch.int64, device=gpu_device)
    log = torch.zeros((4 * 8 * 256, 16), dtype=torch.int64, device=gpu_device)
    assert L.vxrt_render_wave_log(ds.accel, w, h, 0, h, C.byref(p), 0, px.data_ptr(), cnt.data_ptr(), log.data_ptr(), None) == 0
    torch.cuda.synchronize()
    lg = log.cpu().numpy()
    lg = lg[lg[:, 1] > 0]
    # rays started, over the wavefronts of the MAIN launch: every pixel but the column and the row through the middle of the frame (u == 0
    # or v == 0: a zero direction component, traced by the EXACT launch on the side stream, which keeps no log)
    assert len(lg) > 0 and int(lg[:, 2].sum()) == w * h - (w + h - 1)
    xcd = (lg[:, 9].astype(np.uint64) >> np.uint64(56)).astype(np.int64)
    assert xcd.min() >= 0 and xcd.max() <= 7
    busy = lg[lg[:, 3] > 0]                                             # wavefronts that got a tile
    assert len(busy) > 0
    assert (busy[:, 12] > 0).all()                                      # shader clocks of the whole wavefront
    assert (busy[:, 10] > 0).all() and (busy[:, 10] + busy[:, 11] <= busy[:, 12]).all()
    assert (busy[:, 13] + busy[:, 14] <= busy[:, 12]).all()
    assert (busy[:, 4] <= busy[:, 3]).all() and (busy[:, 5] <= 64 * busy[:, 4]).all()
    rpx, _, _ = po.render(sc, w, h)
    assert np.array_equal(px.cpu().numpy().view(np.uint32), rpx)
    ds.close()


@pytest.mark.parametrize("k,shallow", [(5, 1), (16, 1), (17, 0), (32, 0)])
def test_depth_class_picks_the_stack_and_deep_chains_match_the_oracle(vrt, po, gpu_device, k, shallow):
    """The accel build measures the scene's depth (internal levels on the longest root-to-leaf path) and scenes of at most 16 levels
    run with 48-entry stacks, deeper ones with the reference's 32 levels' worth.  A chain-like BVH4 k levels deep whose camera rays
    leave three pending siblings per level (3 k entries at the bottom: exactly 48 at k = 16, 96 at k = 32, the reference's own limit):
    hit records (frame kernels, 7 and 8 wavefronts per SIMD, with shadow rays; ray buffers, closest and any hit) equal the oracle's
    and no overflow is reported."""
    import torch
    from scenes import chain_bvh4
    sc = chain_bvh4(vrt, k)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    assert vrt.rtapi.accel_info(ds.accel, 0) == min(k, 17)
    import os
    assert vrt.rtapi.accel_info(ds.accel, 1) == (0 if os.environ.get("VXRT_SHALLOW") == "0" else shallow)   # (measurement knob: full-size stacks for every scene)
    w, h = 96, 72
    rays = po.camera_rays(w, h)
    want, st = po.trace_canonical(sc, rays)
    assert st["max_stack"] == 3 * k and (want["dist"] < 1e29).mean() > 0.1
    got = gpu_trace(vrt, ds, rays)
    assert np.array_equal(_bits(got), _bits(want))
    wf, _ = po.trace_faithful(sc, rays, any_hit=True)
    assert np.array_equal(_bits(gpu_trace(vrt, ds, rays, mode=vrt.rtapi.MODE_ANY)), _bits(wf))
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = (-50.0, 180.0, 40.0)
    rpx, rhits, _, _ = po.render_ex(sc, w, h, po.shade_params(light_pos=(-50.0, 180.0, 40.0)), 1)
    px, hn, _, _ = gpu_render(vrt, ds, w, h, shadow=1, params=p)
    assert np.array_equal(px, rpx)
    assert np.array_equal(_bits(hn), _bits(rhits))
    ds.close()


def test_a_tree_deeper_than_the_reference_allows_sets_the_overflow_status(vrt, po, gpu_device):
    """36 levels with three pending siblings each need 108 stack entries: more than the reference's 32-level trail allows (undefined
    behaviour there).  The kernels drop what does not fit, flag it (status bit 0) and end; the next scene on the device is unaffected."""
    import torch
    from scenes import chain_bvh4
    sc = chain_bvh4(vrt, 36)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    assert vrt.rtapi.accel_info(ds.accel, 1) == 0
    w, h = 64, 48
    rays = torch.from_numpy(po.camera_rays(w, h)).to(gpu_device)
    out = torch.zeros(w * h * 24, dtype=torch.uint8, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.trace(ds.accel, rays.data_ptr(), w * h, out.data_ptr(), vrt.rtapi.MODE_CLOSEST, None, s)
    assert vrt.rtapi.status(s) & 1
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    vrt.rtapi.render(ds.accel, w, h, 0, h, vrt.rtapi.default_shade_params(), px.data_ptr(), 0, None, None, None, s)
    assert vrt.rtapi.status(s) & 1
    assert vrt.rtapi.status(s) == 0                      # read-and-clear
    ds.close()
    ok = chain_bvh4(vrt, 8)
    d2 = vrt.tracer.DeviceScene(ok, gpu_device)
    got = gpu_trace(vrt, d2, po.camera_rays(w, h))
    assert np.array_equal(_bits(got), _bits(po.trace_canonical(ok, po.camera_rays(w, h))[0]))
    d2.close()
