"""GPU: randomized parity.  Random triangle soups (slivers, zero-area and duplicated triangles among them) as 1-4 instances under
random rotations, non-uniform scales and translations; camera rays, random rays, rays with zero direction components (the EXACT
path) and rays starting inside the geometry; closest hit, first accepted hit, and occlusion with a per-ray bound.  The HIP
traversal must return the oracle's hit records bit for bit -- on the tree the CPU builder makes and on the tree built on the GPU."""
import os

import numpy as np
import pytest

from test_gpu_parity import gpu_trace, _bits

pytestmark = pytest.mark.gpu


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _soup(rng, n):
    c = rng.uniform(-1, 1, size=(n, 1, 3))
    size = rng.choice([0.02, 0.1, 0.4, 1.0], size=(n, 1, 1))
    t = (c + rng.uniform(-1, 1, size=(n, 3, 3)) * size).astype(np.float32)
    t[rng.integers(0, n, max(1, n // 20)), 2] = t[rng.integers(0, n, max(1, n // 20)), 1]      # a few degenerate (and copied) corners
    t[: n // 25] = t[n // 25: 2 * (n // 25)]                                                      # duplicated triangles
    t[rng.integers(0, n, n // 10), :, rng.integers(0, 3)] = np.float32(0.25)                     # axis-aligned flats (zero-thickness boxes)
    return t.reshape(n, 9)


def _rays(rng, po, lo, hi):
    cam = po.camera_rays(48, 32)
    n = 1500
    o = rng.uniform(lo - 20, hi + 20, size=(n, 3))
    tgt = rng.uniform(lo, hi, size=(n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rnd = np.concatenate([o, d], 1)
    inside = np.concatenate([rng.uniform(lo, hi, size=(300, 3)), rng.normal(size=(300, 3))], 1)
    axis = np.zeros((300, 6))
    axis[:, :3] = rng.uniform(lo - 5, hi + 5, size=(300, 3))
    k = rng.integers(0, 3, 300)
    axis[np.arange(300), 3 + k] = rng.choice([-1.0, 1.0], 300)                 # two zero components
    axis[:150, 3 + (k[:150] + 1) % 3] = rng.uniform(-1, 1, 150)              # ... or one
    return np.concatenate([cam, rnd, inside, axis]).astype(np.float32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "6"))))     # (VXRT_FUZZ_SEEDS=n: a soak run over n seeds)
@pytest.mark.parametrize("builder", ["cpu", "gpu"])
def test_random_scenes_and_rays(vrt, po, gpu_device, seed, builder):
    rng = np.random.default_rng(1000 + seed)
    n_inst = int(rng.integers(1, 5))
    meshes, xf = [], []
    for i in range(n_inst):
        meshes.append(_soup(rng, int(rng.integers(40, 400))))
        m = np.eye(4)
        m[:3, :3] = _rot(rng) @ np.diag(rng.uniform(8, 40, 3))
        m[:3, 3] = (rng.uniform(150, 400), rng.uniform(40, 160), rng.uniform(-150, 150))
        xf.append(m.astype(np.float32))
    if builder == "cpu":
        sc = vrt.scene.from_triangles(meshes, xf)
        ds = vrt.tracer.DeviceScene(sc, gpu_device)
    else:
        ds = vrt.tracer.DeviceScene.build_on_gpu(meshes, transforms=xf, device=gpu_device, leaf_max=int(rng.integers(1, 5)))
        sc = ds.to_host()
    lo, hi = np.array([100.0, 0.0, -200.0]), np.array([450.0, 200.0, 200.0])
    rays = _rays(rng, po, lo, hi)
    got = gpu_trace(vrt, ds, rays)
    want, _ = po.trace_canonical(sc, rays)
    assert np.array_equal(_bits(got), _bits(want)), "closest hit"
    if seed < 6:      # (the committed seeds were picked to hit something; a soak run's need not)
        assert (got["dist"] < 1e29).sum() > 100
    got_any = gpu_trace(vrt, ds, rays, mode=1)
    want_any, _ = po.trace_canonical(sc, rays, any_hit=True)
    assert np.array_equal(_bits(got_any), _bits(want_any)), "first accepted hit"
    tmax = rng.uniform(10, 300, len(rays)).astype(np.float32)
    got_t = gpu_trace(vrt, ds, rays, mode=1, tmax=tmax)
    want_t, _ = po.trace_canonical(sc, rays, tmax=tmax, any_hit=True)
    assert np.array_equal(_bits(got_t), _bits(want_t)), "bounded occlusion rays"
    if seed < 6:
        assert 0 < (got_t["dist"] < 1e29).sum() < (got_any["dist"] < 1e29).sum()
    ds.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "4"))))
def test_random_scenes_frames_with_shadow(vrt, po, gpu_device, seed):
    """The frame path on the same kind of scenes: pixels, hit records and the occluded set against the oracle, single frames and a
    batch of three with different lights."""
    import torch
    from test_gpu_parity import gpu_render
    rng = np.random.default_rng(2000 + seed)
    meshes, xf = [], []
    for i in range(int(rng.integers(1, 4))):
        meshes.append(_soup(rng, int(rng.integers(100, 500))))
        m = np.eye(4)
        m[:3, :3] = _rot(rng) @ np.diag(rng.uniform(20, 60, 3))
        m[:3, 3] = (rng.uniform(180, 350), rng.uniform(60, 140), rng.uniform(-100, 100))
        xf.append(m.astype(np.float32))
    sc = vrt.scene.from_triangles(meshes, xf)
    ds = vrt.tracer.DeviceScene(sc, gpu_device)
    w, h = 104, 72
    lights = [(float(rng.uniform(50, 300)), float(rng.uniform(150, 400)), float(rng.uniform(-200, 200))) for _ in range(3)]
    plist = []
    for L in lights:
        p = vrt.rtapi.default_shade_params()
        p.light_pos[:] = L
        plist.append(p)
    px, hits, col, nrays = gpu_render(vrt, ds, w, h, shadow=1, params=plist[0])
    want_px, want_hits, want_col, want_n = po.render_ex(sc, w, h, po.shade_params(light_pos=lights[0]), 1)
    assert np.array_equal(px, want_px) and np.array_equal(_bits(hits.reshape(-1)), _bits(want_hits.reshape(-1))) and nrays == want_n
    assert seed >= 4 or (hits["dist"] < 1e29).mean() > 0.01
    buf = torch.zeros((3, h, w), dtype=torch.int32, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render_batch(ds.accel, w, h, plist, buf.data_ptr(), w * h, 1, None, s)
    torch.cuda.synchronize()
    assert vrt.rtapi.status(s) == 0
    for f in range(3):
        want_f, _, _, _ = po.render_ex(sc, w, h, po.shade_params(light_pos=lights[f]), 1)
        assert np.array_equal(buf[f].cpu().numpy().view(np.uint32), want_f), "frame %d of the batch" % f
    ds.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "4"))))
def test_reference_quirks_mode_equals_the_faithful_restatement_on_random_deep_tlas_scenes(vrt, po, gpu_device, seed):
    """The opt-in quirks traversal (vxrt_trace_reference_quirks) against the oracle's FAITHFUL restatement -- the one pinned to the
    reference's object code -- on random scenes of 6-14 instances (a TLAS deeper than one level, where the stale base_ptr of
    rt_traversal.cpp:91-92 fires) and all ray kinds incl. zero direction components: bit-equal, closest hit and first accepted
    candidate, and the quirk is live (some rays differ from the canonical traversal)."""
    import torch
    rng = np.random.default_rng(7000 + seed)
    n_inst = int(rng.integers(6, 15))
    meshes, xf = [], []
    for i in range(n_inst):
        meshes.append(_soup(rng, int(rng.integers(30, 200))))
        m = np.eye(4)
        m[:3, :3] = _rot(rng) @ np.diag(rng.uniform(8, 40, 3))
        m[:3, 3] = (rng.uniform(150, 400), rng.uniform(40, 160), rng.uniform(-150, 150))
        xf.append(m.astype(np.float32))
    sc = vrt.scene.from_triangles(meshes, xf)
    lo, hi = np.array([100.0, 0.0, -200.0]), np.array([450.0, 200.0, 200.0])
    rays = _rays(rng, po, lo, hi)
    img = po.Image(sc)
    mem = torch.from_numpy(img.mem).to(gpu_device)
    r = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).to(gpu_device)
    offs = (img.off["tlas"], img.off["blas"], img.off["bvh"], img.off["tri"])
    s = torch.cuda.current_stream().cuda_stream
    n = len(rays)
    stale = 0
    for mode, any_hit in ((vrt.rtapi.MODE_CLOSEST, False), (vrt.rtapi.MODE_ANY, True)):
        out = torch.zeros(n * 24, dtype=torch.uint8, device=gpu_device)
        vrt.rtapi.trace_reference_quirks(mem.data_ptr(), mem.numel(), offs, r.data_ptr(), n, out.data_ptr(), mode, None, s)
        assert vrt.rtapi.status(s) == 0
        want, st = po.trace_faithful(img, rays, any_hit=any_hit)
        got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=want.dtype)[:n]
        assert np.array_equal(_bits(got), _bits(want))
        stale += st["stale_base"]
    canon, _ = po.trace_canonical(sc, rays)
    want, _ = po.trace_faithful(img, rays)
    assert seed >= 4 or (stale > 0 and (_bits(canon).reshape(n, -1) != _bits(want).reshape(n, -1)).any(1).sum() > 0)   # (the committed seeds make the quirk live)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "3"))))
def test_random_secondary_ray_passes(vrt, po, gpu_device, seed):
    """The passes built on ray buffers, on random inputs: ambient occlusion (whose any-hit rays take the slot-order traversal since round 5),
    the one-launch diffuse bounce, and the mirror bounce with the shadow extension at every level (a bounce level's occlusion rays take the
    slot-order traversal too) -- mirror hall with random reflectivities, random frame size, light, samples, radius, seed, depth; counts,
    ray totals and pixels equal the oracle's, colours within the colour tolerance."""
    import torch
    from scenes import mirror_hall
    from test_gpu_parity import gpu_render, COLOR_RTOL
    rng = np.random.default_rng(64000 + seed)
    b = mirror_hall(vrt, float(rng.choice([0.0, 0.4, 0.7])), float(rng.choice([0.0, 0.5])))
    ds = vrt.tracer.DeviceScene(b, gpu_device)
    w, h = int(rng.choice([40, 72, 144, 200])), int(rng.choice([24, 61, 88, 120]))
    light = (float(rng.uniform(50, 350)), float(rng.uniform(120, 400)), float(rng.uniform(-200, 200)))
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = light
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    cnt = torch.full((h, w), -1, dtype=torch.int32, device=gpu_device)
    nr = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    # ambient occlusion
    spp, radius, sd = int(rng.choice([1, 3, 8, 16])), float(rng.uniform(5, 120)), int(rng.integers(1, 1 << 30))
    vrt.rtapi.render_ao(ds.accel, w, h, 0, h, p, spp, radius, px.data_ptr(), seed=sd, colors_ptr=col.data_ptr(), unoccluded_ptr=cnt.data_ptr(), rays_ptr=nr.data_ptr(), stream=s)
    assert vrt.rtapi.status(s) == 0
    rpx, rcol, rcnt, rn = po.render_ao(b, w, h, po.shade_params(light_pos=light), spp=spp, radius=radius, seed=sd)
    np.testing.assert_array_equal(cnt.cpu().numpy().view(np.uint32), rcnt)
    assert int(nr.item()) == rn
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), rpx)
    # one diffuse bounce
    nr.zero_()
    vrt.rtapi.render_diffuse_bounce(ds.accel, w, h, 0, h, p, px.data_ptr(), seed=sd, colors_ptr=col.data_ptr(), rays_ptr=nr.data_ptr(), stream=s)
    assert vrt.rtapi.status(s) == 0
    rpx, rcol, rn = po.render_gi(b, w, h, po.shade_params(light_pos=light), seed=sd)
    assert int(nr.item()) == rn
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), rpx)
    # mirror bounce, occlusion rays at every level
    depth, shadow = int(rng.integers(1, 6)), int(rng.integers(0, 2))
    p.max_depth = depth
    gpx, ghits, gcol, nrays = gpu_render(vrt, ds, w, h, shadow=shadow, params=p)
    rpx, rhits, rcol, rn = po.render_ex(b, w, h, po.shade_params(light_pos=light, max_depth=depth), shadow)
    assert np.array_equal(_bits(ghits), _bits(rhits)) and nrays == rn
    np.testing.assert_allclose(gcol, rcol, rtol=COLOR_RTOL)
    np.testing.assert_array_equal(gpx, rpx)
    ds.close()
