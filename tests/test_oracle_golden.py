"""CPU: the oracle restatement against the committed fixtures produced by the reference's own object
code (oracle/gen_golden.py).  This is what pins the oracle when /root/reference is absent."""
import numpy as np
import pytest

# tex_mix: textured materials (uv outside [0,1], two non-power-of-two PNGs of the reference) next to an untextured one
SCENES = ["teapot", "torus", "sphere", "cone", "cylinder", "cube", "teapot_x3", "sphere_x6", "tex_mix"]
# TLAS deeper than one level: the reference traverser addresses the children of a TLAS internal node
# popped from its short stack relative to the last BLAS's base (rt_traversal.cpp:91-92 vs :119-120),
# reads unrelated nodes and loses real hits.  The faithful restatement reproduces that bit for bit;
# the canonical algorithm (and the HIP kernels) address TLAS children from the TLAS base.  Rays that
# trip the quirk are excluded from canonical-vs-reference comparison and counted instead.
QUIRK_SCENES = {"sphere_x6"}


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.parametrize("name", SCENES)
def test_faithful_restatement_matches_reference_hits(po, golden, name):
    g = golden(name)
    hits, st = po.trace_faithful(g, g["rays"])
    assert np.array_equal(_bits(hits), _bits(g["hits"])), "closest-hit records differ from the reference traverser"
    # the restatement follows the reference node for node: same number of node / triangle fetches
    assert st["node_reads"] == int(g["ref_node_reads"])
    assert st["tri_reads"] == int(g["ref_tri_reads"])
    assert st["trail_overflow"] == 0 and st["oob"] == 0


@pytest.mark.parametrize("name", SCENES)
def test_canonical_algorithm_matches_reference_hits(po, golden, name):
    """The single-pass full-stack algorithm the HIP kernels implement returns the reference's hit
    (distance bits, barycentrics, blasIdx, triIdx), exact-distance ties included."""
    g = golden(name)
    hits, st = po.trace_canonical(g, g["rays"])
    ok = ~po.stale_base_mask(g, g["rays"]) if name in QUIRK_SCENES else np.ones(len(hits), bool)
    assert np.array_equal(_bits(hits[ok]), _bits(g["hits"][ok]))


@pytest.mark.parametrize("name", SCENES)
def test_any_hit_first_matches_reference(po, golden, name):
    g = golden(name)
    f, _ = po.trace_faithful(g, g["rays"], any_hit=True)
    c, _ = po.trace_canonical(g, g["rays"], any_hit=True)
    assert np.array_equal(_bits(f), _bits(g["anyhits"]))
    ok = ~po.stale_base_mask(g, g["rays"]) if name in QUIRK_SCENES else np.ones(len(c), bool)
    assert np.array_equal(_bits(c[ok]), _bits(g["anyhits"][ok]))


@pytest.mark.parametrize("name", SCENES)
def test_shading_matches_reference_helpers(po, golden, name):
    g = golden(name)
    col, px = po.shade(g, g["rays"], g["hits"])
    assert np.array_equal(_bits(col), _bits(g["colors"])), "f32 radiance differs from closest.cpp/rtx_shading.h"
    assert np.array_equal(px, g["rgb8"])


def test_camera_rays_match_reference_kernel(po, golden):
    rows = golden("camera_rays")["rows"]
    import ctypes as C
    L = po.orc()
    tmp = (C.c_float * 6)()
    for x, y, w, h, *bits in rows[:: max(1, len(rows) // 1500)]:
        L.orc_generate_ray(int(x), int(y), int(w), int(h), tmp)
        got = np.array(tmp[:], np.float32).view(np.uint32)
        assert np.array_equal(got, np.array(bits, np.uint32)), (x, y, w, h)


def test_exact_distance_ties_are_present_and_resolved_like_the_reference(po, golden):
    """on the cube two triangles of one face report bit-identical distances; the winner
    is decided by traversal order, not by lowest index.  Count the ties so they are never silently
    tolerated."""
    g = golden("cube")
    tri = g["tri"].view(np.float32).reshape(-1, 9)
    import ctypes as C
    L = po.orc()
    L.orc_ray_tri.restype = C.c_float
    L.orc_ray_tri.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ties = 0
    differs_from_lowest_index = 0
    for r, h in zip(g["rays"], g["hits"]):
        if h["dist"] >= 1e29:
            continue
        same = []
        for t in range(len(tri)):
            b = (C.c_float * 3)()
            d = L.orc_ray_tri(r.ctypes.data, tri[t].ctypes.data, C.byref(b, 0), C.byref(b, 4), C.byref(b, 8))
            if np.float32(d) == h["dist"]:
                same.append(t)
        assert int(h["triIdx"]) in same
        if len(same) > 1:
            ties += 1
            differs_from_lowest_index += int(h["triIdx"]) != same[0]
    assert ties > 0, "fixture no longer exercises exact-distance ties"
    print("cube fixture: %d tie rays, %d resolved differently from lowest-index" % (ties, differs_from_lowest_index))


def test_stale_base_quirk_is_confined_to_deep_tlas(po, golden):
    """One-level TLAS (<= 4 instances under the root) never trips the quirk; the 6-instance fixture
    does, and there the reference loses hits that exist (brute force over all instances finds them)."""
    _, st = po.trace_faithful(golden("teapot_x3"), golden("teapot_x3")["rays"])
    assert st["stale_base"] == 0
    g = golden("sphere_x6")
    mask = po.stale_base_mask(g, g["rays"])
    assert int(mask.sum()) == 43 and len(mask) == 1088        # a fixed set (the GPU tests exclude exactly these rays)
    c, _ = po.trace_canonical(g, g["rays"])
    lost = (g["hits"]["dist"][mask] >= 1e29) & (c["dist"][mask] < 1e29)
    print("sphere_x6: %d rays trip the quirk, reference loses a real hit on %d of them" % (mask.sum(), lost.sum()))
    assert (c["dist"][mask] <= g["hits"]["dist"][mask]).all() and int(lost.sum()) == 21


def test_mirror_bounce_restatement_properties(vrt, po):
    """orc_render_ex (closest.cpp:95-121 followed recursively; unpinned arm, see oracle/README.md):
    depth 1 is orc_render; without reflective instances depth does not matter; with them, pixels that
    see a mirror change and each level only adds rays."""
    from scenes import mirror_hall
    w, h = 64, 40
    lp = (150.0, 220.0, -60.0)
    b = mirror_hall(vrt)
    px1, h1, c1 = po.render(b, w, h, po.shade_params(light_pos=lp, max_depth=1))
    pxe, he, ce, n1 = po.render_ex(b, w, h, po.shade_params(light_pos=lp, max_depth=1))
    assert np.array_equal(px1, pxe) and np.array_equal(c1.view(np.uint32), ce.view(np.uint32)) and n1 == w * h
    dull = mirror_hall(vrt, 0.0, 0.0)
    a = po.render_ex(dull, w, h, po.shade_params(light_pos=lp, max_depth=1))
    d = po.render_ex(dull, w, h, po.shade_params(light_pos=lp, max_depth=5))
    assert np.array_equal(a[2].view(np.uint32), d[2].view(np.uint32)) and a[3] == d[3]
    counts = [po.render_ex(b, w, h, po.shade_params(light_pos=lp, max_depth=k))[3] for k in (1, 2, 3, 4)]
    assert counts[0] < counts[1] < counts[2] <= counts[3]
    _, hits, c4, _ = po.render_ex(b, w, h, po.shade_params(light_pos=lp, max_depth=4))
    sees_mirror = (hits["dist"] < 1e29) & (hits["blasIdx"] == 1)
    assert sees_mirror.any()
    changed = (c4.view(np.uint32) != c1.view(np.uint32)).any(-1)
    # (a mirror ray that leaves the scene returns the background, which is what the else arm adds too)
    assert changed[sees_mirror].any() and not changed[~sees_mirror].any()
    # one level by hand: C = term + C(mirror ray) * reflectivity, with term = shade(reflectivity -> the else arm) - bg * refl
    y, x = np.argwhere(sees_mirror & changed)[0]
    rays = po.camera_rays(w, h)[y * w + x][None]
    p2 = po.shade_params(light_pos=lp, max_depth=2)
    c2 = po.render_ex(b, w, h, p2)[2][y, x]
    f = np.float32
    refl = f(b["blas"].view(np.float32).reshape(-1, 40)[1, 38])
    bg = np.array(p2.background[:], f)
    else_arm = po.shade(b, rays, hits[y, x][None], p2)[0][0]               # term + bg * refl
    n_hat = np.array([-1.0, 0.0, 0.0], f)                                   # mirror 1 faces -x
    dvec = rays[0, 3:].astype(f)
    R = dvec - (f(2.0) * n_hat) * f(np.dot(n_hat, dvec))
    R = (R * (f(1.0) / np.sqrt(f(np.dot(R, R)), dtype=f))).astype(f)
    I = (rays[0, :3] + dvec * hits[y, x]["dist"]).astype(f)
    sec = np.concatenate([I + R * f(0.001), R]).astype(f)[None]
    sh, _ = po.trace_faithful(b, sec)
    csec = po.shade(b, sec, sh, po.shade_params(light_pos=lp, max_depth=1))[0][0]
    want = (else_arm - bg * refl) + csec * refl
    np.testing.assert_allclose(c2, want, rtol=2e-6, atol=1e-7)


def test_rng_matches_reference_fixture(po, golden):
    """WangHash / RandomInt / RandomFloat of the restatement (what the AO and diffuse-bounce passes draw from) == the reference's own
    helpers (common.h:129-147, host-compiled in oracle/_ref, tests/golden/rng.npz): 7 seeds x 256 values, bit for bit."""
    g = golden("rng")
    for k, seed in enumerate(g["seeds"]):
        h, i, f = po.rng(int(seed), 256)
        assert np.array_equal(h, g["hash%d" % k]) and np.array_equal(i, g["ints%d" % k])
        assert np.array_equal(f.view(np.uint32), g["floats%d" % k].view(np.uint32))
    assert len(np.unique(g["hash0"])) == 256 and (g["floats3"] >= 0).all() and (g["floats3"] <= 1).all()


MIRROR_FIXTURES = ("mirror_teapot", "mirror_trio")


def _mirror_params(po, g, depth):
    L = g["light12"]
    return po.shade_params(ambient=tuple(L[6:9]), light_color=tuple(L[3:6]), light_pos=tuple(L[0:3]), background=tuple(L[9:12]), max_depth=depth)


@pytest.mark.parametrize("name", MIRROR_FIXTURES)
def test_mirror_arm_matches_the_reference_twin_fixture(po, golden, name):
    """The mirror arm (shaders/closest.cpp:95-121) pinned to reference OBJECT CODE: the fixture holds what the reference's software twin
    (raycast/render.h:210-277, compiled where it lies; oracle/gen_golden_mirror.py) returns for the RTU kernel's own camera rays on
    a scene both reference builders built, reflective instances 0.5 / 0.3 as raycast/tracer.cpp:13, at max_depth 1..4.  The
    restatement of the RTU shader recursion on the RTU buffers gives the same radiance within 1e-5 relative (measured: 1.4e-7 -- the
    twin carries a running throughput, the RTU shader multiplies on the way back) and the same RGB8 in every pixel; at depth 1 the two
    are bit-equal.  The bounce does something: dozens of pixels change from each depth to the next."""
    g = golden(name)
    w, h = int(g["width"]), int(g["height"])
    prev = None
    for d in [int(x) for x in g["depths"]]:
        px, hits, col, n = po.render_ex(g, w, h, _mirror_params(po, g, d), 0)
        want, wpx = g["colors_d%d" % d], g["rgb8_d%d" % d]
        np.testing.assert_allclose(col, want, rtol=1e-5, atol=0)
        assert np.abs(col - want).max() <= 2e-6 * np.abs(want).max()
        assert np.array_equal(px, wpx), "%d RGB8 pixels differ at depth %d" % (int((px != wpx).sum()), d)
        if d == 1:
            assert np.array_equal(col.view(np.uint32), want.view(np.uint32))
        if prev is not None:
            assert int((wpx != prev).sum()) >= 8, "the bounce at depth %d changes no pixel" % d
        prev = wpx
    assert float(g["hit_fraction"]) > 0.05 and (g["reflectivity"] > 0).any()
