"""Software twin (tests/regression/raycast).  The fixtures tests/golden/rc_*.npz hold buffers built
by the reference's own raycast Scene/BVH/TLAS code and what its own CPU path (GenerateRay/Trace, tracer.cpp:249-263)
rendered: the restatement oracle/rc_oracle.c and the HIP kernel must reproduce the pixels exactly."""
import numpy as np
import pytest

RC_FIXTURES = ["rc_teapot", "rc_teapot_x3", "rc_torus_x2", "rc_cube_x2"]


def _args(po, g):
    sc = {k: g[k] for k in po.RC_BUFFERS}
    sc["tlas_root"] = int(g["tlas_root"])
    return po.rc_args(sc, int(g["width"]), int(g["height"]), g["cam14"], g["light12"], int(g["spp"]), int(g["max_depth"]))


@pytest.mark.parametrize("name", RC_FIXTURES)
def test_restatement_matches_reference_cpu_path(po, golden, name):
    g = golden(name)
    a = _args(po, g)
    px, col = po.rc_render(a)
    np.testing.assert_array_equal(px, g["pixels"])
    assert (px != px[0, 0]).mean() > 0.1
    hits = po.rc_trace(a, g["rays"])
    assert np.array_equal(hits.view(np.uint8), g["hits"].view(np.uint8))
    np.testing.assert_array_equal(po.rc_camera_rays(a)[::5], g["rays"])


@pytest.mark.ref
@pytest.mark.parametrize("vfov,zoom,depth,spp", [(45.0, 1.0, 1, 1), (46.0, 1.0, 4, 2)])
def test_restatement_equals_reference_object_code_live(po, golden, vfov, zoom, depth, spp):
    """Fresh cameras on reference-built scenes, through oracle/_ref/libvxref_rc.so (skipped where it was not built)."""
    import os
    A = "/root/reference/tests/regression/raycast/assets/"
    if not po.have_ref_rc() or not os.path.isdir(A):
        pytest.skip("oracle/_ref/libvxref_rc.so or the reference assets are not here")
    sc = po.RefRcScene([A + "sphere.obj", A + "cone.obj"], [A + "bricks.png", A + "green.png"], [0.35, 0.0])
    w, h = 72, 52
    cam = sc.camera(vfov, zoom, w, h)
    ref = sc.render(w, h, spp, depth, cam, po.RC_DEFAULT_LIGHT)
    b = dict(sc.buffers)
    b["tlas_root"] = sc.tlas_root
    px, _ = po.rc_render(po.rc_args(b, w, h, cam, po.RC_DEFAULT_LIGHT, spp, depth))
    sc.close()
    np.testing.assert_array_equal(px, ref)
    assert (ref != ref[0, 0]).mean() > 0.01


def test_png_decoder_equals_stb_on_a_reference_asset(vrt, golden, tmp_path):
    """rc_teapot carries the reference's red.png (data file) and, in its texture buffer, the texels stb_image decoded
    from it (surface.cpp:28-55): the package's own PNG decoder must produce the same words."""
    g = golden("rc_teapot")
    f = tmp_path / "red.png"
    f.write_bytes(g["red_png"].tobytes())
    got = vrt.scene.image_load(f)
    blas = g["blas"].view(np.uint32).reshape(-1, 40)
    tw, th, off = int(blas[0, 36]), int(blas[0, 37]), int(blas[0, 34])
    assert got.shape == (th, tw)
    want = g["tex"][off:off + tw * th * 4].view(np.uint32).reshape(th, tw)
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RC_FIXTURES)
def test_hip_twin_matches_reference_pixels(vrt, po, golden, gpu_device, name):
    import torch
    g = golden(name)
    sc = {k: g[k] for k in po.RC_BUFFERS}
    sc["tlas_root"] = int(g["tlas_root"])
    w, h = int(g["width"]), int(g["height"])
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    prm = vrt.rtapi.rc_params(g["cam14"], g["light12"], int(g["spp"]), int(g["max_depth"]))
    px = torch.full((h, w), -1, dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    stream = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.rc_render(ds.c, w, h, 0, h, prm, px.data_ptr(), col.data_ptr(), stream)
    assert vrt.rtapi.status(stream) == 0
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), g["pixels"])
    _, ocol = po.rc_render(_args(po, g))
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), ocol, rtol=1e-5)
    # row window: only rows [y0,y1) are written
    px2 = torch.full((h, w), 0x7EADBEEF, dtype=torch.int32, device=gpu_device)
    vrt.rtapi.rc_render(ds.c, w, h, 8, 20, prm, px2.data_ptr(), None, stream)
    out = px2.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(out[8:20], g["pixels"][8:20])
    assert (out[:8] == 0x7EADBEEF).all() and (out[20:] == 0x7EADBEEF).all()


@pytest.mark.gpu
def test_twin_through_vx_api_and_bad_scenes(vrt, po, golden, gpu_device):
    """raycast/tracer.cpp's call sequence (vx_mem_alloc x8, vx_copy_to_dev x7, vx_upload_bytes(kernel_arg_t 192 B),
    vx_start, vx_ready_wait, vx_copy_from_dev) against libvortex-hip.so with the raycast kernel selector."""
    g = golden("rc_teapot_x3")
    sc = {k: g[k] for k in po.RC_BUFFERS}
    sc["tlas_root"] = int(g["tlas_root"])
    w, h = int(g["width"]), int(g["height"])
    tr = vrt.tracer.RaycastTracer(w, h, int(g["spp"]), int(g["max_depth"]))
    tr.init(sc)
    tr.setup(g["cam14"], g["light12"])
    px = tr.run()
    np.testing.assert_array_equal(px, g["pixels"])
    # a BVH node pointing outside the buffer is caught by the in-kernel bounds checks: ready_wait fails, nothing faults
    bad = {k: v.copy() for k, v in sc.items() if k != "tlas_root"}
    bad["tlas_root"] = sc["tlas_root"]
    nodes = bad["bvh"].view(np.uint32).reshape(-1, 8)
    inner = int(np.nonzero(nodes[:, 7] == 0)[0][0])
    nodes[inner, 3] = 0x7FFFFFF0
    tr2 = vrt.tracer.RaycastTracer(w, h, 1, 1)
    tr2.init(bad)
    tr2.setup(g["cam14"], g["light12"])
    with pytest.raises(vrt.runtime.VxError):
        tr2.run()
    tr.close()


def test_rc_scene_builder_invariants_and_oracle_render(vrt, po):
    """The package's BVH2/TLAS builder in the twin's formats: children adjacent, leaves index triIdx, boxes contain
    their triangles, TLAS child indices 16 bit, and the restatement renders it (hits = brute force over all triangles)."""
    sc = vrt.scene.rc_procedural("blob", 3, 0, 2, copies=3, reflectivity=[0.0, 0.4, 0.0])
    bvh = sc["bvh"].view(np.uint32).reshape(-1, 8)
    fb = sc["bvh"].view(np.float32).reshape(-1, 8)
    tri = sc["tri"].view(np.float32).reshape(-1, 9)
    idx = sc["triIdx"].view(np.uint32)
    blas = sc["blas"].view(np.uint32).reshape(-1, 40)
    assert len(blas) == 3 and len(idx) == len(tri)
    covered = np.zeros(len(tri), int)
    for b in range(3):
        base = int(blas[b, 32])
        end = int(blas[b + 1, 32]) if b + 1 < 3 else len(bvh)
        stack = [0]
        while stack:
            n = stack.pop()
            lf, tc = int(bvh[base + n, 3]), int(bvh[base + n, 7])
            lo, hi = fb[base + n, 0:3], fb[base + n, 4:7]
            if tc:
                t = tri[idx[lf:lf + tc]].reshape(-1, 3)
                assert (t >= lo - 1e-4).all() and (t <= hi + 1e-4).all()
                covered[idx[lf:lf + tc]] += 1
            else:
                assert 0 < lf and lf + 1 < end - base
                for c in (lf, lf + 1):
                    assert (fb[base + c, 0:3] >= lo - 1e-4).all() and (fb[base + c, 4:7] <= hi + 1e-4).all()
                stack += [lf, lf + 1]
    assert (covered == 1).all()
    tl = sc["tlas"].view(np.uint32).reshape(-1, 8)
    assert tl[sc["tlas_root"], 3] != 0 and (tl[1:, 3] >> 16 < len(tl)).all()
    w, h = 64, 40
    cam = vrt.scene.rc_camera_like_rtu(w, h)
    a = po.rc_args(sc, w, h, cam, po.RC_DEFAULT_LIGHT, 1, 2)
    px, col = po.rc_render(a)
    rays = po.rc_camera_rays(a)
    hits = po.rc_trace(a, rays)
    assert 0.02 < (hits["dist"] < 1e29).mean() < 0.98
    # brute force in instance space for a sample of rays
    inv = sc["blas"].view(np.float32).reshape(-1, 40)[:, 16:32].reshape(-1, 4, 4)
    counts = np.diff(np.append(blas[:, 32], len(bvh)))
    tri_of = np.cumsum([0] + [len(tri) // 3] * 3)
    for r in rays[:: max(1, len(rays) // 60)]:
        best = 1e30
        for b in range(3):
            o = inv[b][:3, :3] @ r[:3] + inv[b][:3, 3]
            d = inv[b][:3, :3] @ r[3:]
            T = tri[tri_of[b]:tri_of[b + 1]].astype(np.float64)
            v0, e1, e2 = T[:, 0:3], T[:, 3:6] - T[:, 0:3], T[:, 6:9] - T[:, 0:3]
            hh = np.cross(d, e2); aa = (e1 * hh).sum(1)
            ok = np.abs(aa) >= 1e-6
            f = np.where(ok, 1.0 / np.where(ok, aa, 1), 0)
            s_ = o - v0; u = f * (s_ * hh).sum(1); q = np.cross(s_, e1); v = f * (q @ d); t = f * (e2 * q).sum(1)
            m = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-6)
            if m.any():
                best = min(best, t[m].min())
        got = float(po.rc_trace(a, r[None])["dist"][0])
        assert (best > 1e29 and got > 1e29) or abs(got - best) <= 2e-4 * max(1.0, best)


@pytest.mark.gpu
def test_hip_twin_on_package_built_scene(vrt, po, gpu_device):
    import torch
    sc = vrt.scene.rc_procedural("blob", 4, 0, 5, copies=3, reflectivity=[0.5, 0.0, 0.3])
    w, h = 200, 120
    cam = vrt.scene.rc_camera_like_rtu(w, h)
    light = (150.0, 300.0, -80.0, 1, 1, 1, 0.3, 0.3, 0.3, 0.4, 0.35, 0.25)
    opx, ocol = po.rc_render(po.rc_args(sc, w, h, cam, light, 2, 3))
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.rc_render(ds.c, w, h, 0, h, vrt.rtapi.rc_params(cam, light, 2, 3), px.data_ptr(), col.data_ptr(), s)
    assert vrt.rtapi.status(s) == 0
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), opx)
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), ocol, rtol=1e-5)
    assert (opx != opx[0, 0]).mean() > 0.05


@pytest.mark.gpu
def test_twin_accel_is_reusable_and_rejects_malformed_scenes(vrt, po, golden, gpu_device):
    """vxrc_accel_build once, several frames on it (different samples / depth / row windows) equal the restatement; a scene whose
    BVH walk would leave its buffers fails at build time (-1) instead of reaching a kernel."""
    import torch
    g = golden("rc_teapot_x3")
    sc = {k: g[k] for k in po.RC_BUFFERS}
    sc["tlas_root"] = int(g["tlas_root"])
    w, h = int(g["width"]), int(g["height"])
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    for spp, depth in ((1, 1), (int(g["spp"]), int(g["max_depth"])), (3, 5)):
        prm = vrt.rtapi.rc_params(g["cam14"], g["light12"], spp, depth)
        px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
        vrt.rtapi.rc_render_accel(ds.accel, w, h, 0, h, prm, px.data_ptr(), None, s)
        assert vrt.rtapi.status(s) == 0
        want, _ = po.rc_render(po.rc_args(sc, w, h, g["cam14"], g["light12"], spp, depth))
        np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), want)
    ds.close()

    def bad_scene(mut):
        b = {k: v.copy() for k, v in sc.items() if k != "tlas_root"}
        b["tlas_root"] = sc["tlas_root"]
        mut(b)
        d2 = vrt.tracer.RcDeviceScene(b, gpu_device)
        with pytest.raises(vrt.runtime.VxError):
            d2.accel
        d2.close()

    nodes = sc["bvh"].view(np.uint32).reshape(-1, 8)
    inner = int(np.nonzero(nodes[:, 7] == 0)[0][0])
    inner2 = int(np.nonzero((nodes[:, 7] == 0) & (np.arange(len(nodes)) > 6))[0][0])
    leaf = int(np.nonzero(nodes[:, 7] != 0)[0][0])

    def child_out_of_range(b):
        b["bvh"].view(np.uint32).reshape(-1, 8)[inner, 3] = 0x7FFFFFF0

    def child_before_parent(b):       # would be a cycle
        b["bvh"].view(np.uint32).reshape(-1, 8)[inner2, 3] = 0

    def leaf_past_tri_idx(b):
        b["bvh"].view(np.uint32).reshape(-1, 8)[leaf, 3] = len(b["triIdx"]) // 4

    def tri_idx_past_tris(b):
        b["triIdx"].view(np.uint32)[5] = 0x00FFFFFF

    def bad_bvh_offset(b):
        b["blas"].view(np.uint32).reshape(-1, 40)[0, 32] = 0x7FFFFFFF

    for mut in (child_out_of_range, child_before_parent, leaf_past_tri_idx, tri_idx_past_tris, bad_bvh_offset):
        bad_scene(mut)
    torch.cuda.synchronize()


def _chain_scene(vrt, k):
    """A chain-like BVH2, 2k internal levels deep, in the twin's formats: node N_i = (A_i = (leaf, leaf), B_i = (leaf, N_i+1)), triangles
    stacked along the view direction, larger and farther with depth, so that every ray hits every box and the walk -- which visits the
    FARTHER child first (render.h:110) -- goes all the way down with everything else left on its stack: one entry per level in the
    reference's walk, three per two levels in a walk that takes two levels per fetch."""
    sc = vrt.scene.rc_procedural("cornell")
    tris, nodes = [], []

    def tri(i):
        x, s = 200.0 + 4.0 * i, 12.0 + 2.5 * i
        tris.append([x, 100.0 - s, -s, x, 100.0 + s, -s, x, 100.0 + 0.3 * s, s])
        return len(tris) - 1

    def leaf(t):
        v = np.array(tris[t], np.float32).reshape(3, 3)
        return {"lo": v.min(0), "hi": v.max(0), "first": t, "count": 1}

    def union(a, b):
        return {"lo": np.minimum(a["lo"], b["lo"]), "hi": np.maximum(a["hi"], b["hi"]), "kids": (a, b)}

    def build(i, left):
        if left == 0:
            return leaf(tri(3 * i))
        a = union(leaf(tri(3 * i)), leaf(tri(3 * i + 1)))
        b1 = leaf(tri(3 * i + 2))
        return union(a, union(b1, build(i + 1, left - 1)))

    root = build(0, k)
    out = [None, None]                       # slot 1 stays empty, as in the reference's 2N-node buffer (children are adjacent pairs)
    def emit(n, at):
        out[at] = n
        if "kids" in n:
            l = len(out)
            out.extend([None, None])
            n["left"] = l
            emit(n["kids"][0], l)
            emit(n["kids"][1], l + 1)
    import sys
    sys.setrecursionlimit(10000)
    emit(root, 0)
    bvh = np.zeros((len(out), 8), np.uint32)
    fb = bvh.view(np.float32)
    for i, n in enumerate(out):
        if n is None:
            continue
        fb[i, 0:3], fb[i, 4:7] = n["lo"], n["hi"]
        if "kids" in n:
            bvh[i, 3], bvh[i, 7] = n["left"], 0
        else:
            bvh[i, 3], bvh[i, 7] = n["first"], n["count"]
    nt = len(tris)
    ex = np.zeros((nt, 15), np.float32)
    ex[:, 0] = ex[:, 3] = ex[:, 6] = -1.0
    b = {k_: v.copy() for k_, v in sc.items() if k_ in ("tlas", "blas", "tex")}
    b["bvh"] = bvh.view(np.uint8).reshape(-1)
    b["tri"] = np.array(tris, np.float32).view(np.uint8).reshape(-1)
    b["triEx"] = ex.view(np.uint8).reshape(-1)
    b["triIdx"] = np.arange(nt, dtype=np.uint32).view(np.uint8)
    b["tlas_root"] = int(sc["tlas_root"])
    tl = b["tlas"].view(np.float32).reshape(-1, 8)
    tl[:, 0:3], tl[:, 4:7] = root["lo"], root["hi"]
    assert len(b["blas"]) == 160 and b["blas"].view(np.uint32)[32] == 0
    return b


def test_restatement_walks_a_chain_of_fifty_levels(vrt, po):
    """The reference's own walk needs one stack entry per level: 50 < BVH_STACK_SIZE = 64, so the restatement renders the chain."""
    sc = _chain_scene(vrt, 25)
    w, h = 48, 32
    px, _ = po.rc_render(po.rc_args(sc, w, h, vrt.scene.rc_camera_like_rtu(w, h), po.RC_DEFAULT_LIGHT, 1, 1))
    assert len(np.unique(px)) > 3


@pytest.mark.gpu
@pytest.mark.parametrize("k,wide", [(21, 1), (22, 0), (25, 0), (31, 0)])
def test_hip_twin_on_a_chain_like_tree(vrt, po, gpu_device, k, wide):
    """A BVH2 2k internal levels deep that the reference walks inside its 64 stack entries.  The walk that takes two levels per fetch
    leaves up to three entries per two levels: it is used up to 42 levels (63 entries) and the build keeps the reference's own walk
    beyond -- either way the frame is the restatement's, and no stack overflow is reported."""
    import torch
    sc = _chain_scene(vrt, k)
    w, h = 96, 64
    cam = vrt.scene.rc_camera_like_rtu(w, h)
    want, _ = po.rc_render(po.rc_args(sc, w, h, cam, po.RC_DEFAULT_LIGHT, 1, 1))
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    assert vrt.rtapi.rc_accel_info(ds.accel, 1) == min(2 * k, 43)     # (counted up to the first level the wide walk could not hold)
    assert vrt.rtapi.rc_accel_info(ds.accel, 0) == wide
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.rc_render_accel(ds.accel, w, h, 0, h, vrt.rtapi.rc_params(cam, po.RC_DEFAULT_LIGHT, 1, 1), px.data_ptr(), None, s)
    assert vrt.rtapi.status(s) == 0
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), want)
    assert len(np.unique(want)) > 3
    ds.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VXRT_FUZZ_SEEDS", "4"))))     # (VXRT_FUZZ_SEEDS=n: a soak run over n seeds)
def test_hip_twin_on_random_package_built_scenes(vrt, po, gpu_device, seed):
    """The twin's kernel against the restatement of render.h on scenes the package builds in the twin's formats: 1-4 instances of a blob
    of 80-1,280 triangles with random reflectivities, random light, 1-3 bounces, 1-2 samples -- pixels equal, colours to 1e-5."""
    import torch
    rng = np.random.default_rng(31000 + seed)
    copies = int(rng.integers(1, 5))
    sc = vrt.scene.rc_procedural("blob", int(rng.integers(2, 5)), 0, int(rng.integers(1, 1000)), copies=copies,
                                 reflectivity=[float(r) for r in rng.choice([0.0, 0.0, 0.3, 0.6], size=copies)])
    w, h = int(rng.choice([64, 136, 200])), int(rng.choice([40, 61, 120]))
    cam = vrt.scene.rc_camera_like_rtu(w, h)
    light = (float(rng.uniform(-200, 400)), float(rng.uniform(100, 500)), float(rng.uniform(-200, 200)), 1, 1, 1, 0.3, 0.3, 0.3, 0.4, 0.35, 0.25)
    depth, spp = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    opx, ocol = po.rc_render(po.rc_args(sc, w, h, cam, light, spp, depth))
    ds = vrt.tracer.RcDeviceScene(sc, gpu_device)
    px = torch.zeros((h, w), dtype=torch.int32, device=gpu_device)
    col = torch.zeros(h * w * 3, dtype=torch.float32, device=gpu_device)
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.rc_render(ds.c, w, h, 0, h, vrt.rtapi.rc_params(cam, light, spp, depth), px.data_ptr(), col.data_ptr(), s)
    assert vrt.rtapi.status(s) == 0
    np.testing.assert_array_equal(px.cpu().numpy().view(np.uint32), opx)
    np.testing.assert_allclose(col.cpu().numpy().reshape(h, w, 3), ocol, rtol=1e-5)
    ds.close() if hasattr(ds, "close") else None
