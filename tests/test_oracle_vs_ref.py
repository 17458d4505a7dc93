"""Live pin of the restatement: oracle/rt_oracle.c against the reference's own object code
(oracle/_ref/libvxref.so = sim/simx/rt_traversal.cpp compiled where it lies) on rays the committed
fixtures do not contain.  Skipped where the library was not built (no /root/reference at build time)."""
import numpy as np
import pytest

pytestmark = pytest.mark.ref


def _fresh_rays(g, n, seed):
    """Rays aimed at the scene from outside, seeded; not the fixture's rays."""
    rng = np.random.default_rng(seed)
    tri = g["tri"].view(np.float32).reshape(-1, 3)
    lo, hi = tri.min(0), tri.max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = c + d * (2.5 * r)
    t = c + rng.normal(size=(n, 3)) * (0.35 * r)
    v = t - o
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.concatenate([o, v], 1).astype(np.float32)


@pytest.mark.parametrize("name,n", [("teapot", 3000), ("torus", 3000), ("sphere", 3000), ("teapot_x3", 2000)])
def test_restatement_equals_reference_object_code(po, golden, name, n):
    if not po.have_ref():
        pytest.skip("oracle/_ref/libvxref.so not built")
    g = golden(name)
    rays = _fresh_rays(g, n, seed=20261004)
    ref_hits, ref_st = po.trace_ref(g, rays)
    got, st = po.trace_faithful(g, rays)
    assert (ref_hits["dist"] < 1e29).sum() > n // 10
    assert np.array_equal(got.view(np.uint8), ref_hits.view(np.uint8))
    can, _ = po.trace_canonical(g, rays)
    assert np.array_equal(can.view(np.uint8), ref_hits.view(np.uint8))
    ref_any, _ = po.trace_ref(g, rays, any_hit=True)
    got_any, _ = po.trace_faithful(g, rays, any_hit=True)
    assert np.array_equal(got_any.view(np.uint8), ref_any.view(np.uint8))


def test_obj_ingest_equals_the_reference_mesh_loader(vrt, po, tmp_path):
    """the package's own OBJ / MTL / PNG readers against the reference's Mesh loader (tinyobj + stb_image, compiled
    where they lie in oracle/_ref) on the same files: the same triangles with the same normals / uvs / material ids (as a
    multiset -- each builder reorders triangles for its own tree), the same materials (the fields the reference initialises:
    it leaves `illum`, and `tex_offset` of an untextured material, uninitialised), the same texels."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libvxref.so not built")
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_reference_host import _write_obj
    d = str(tmp_path)
    _write_obj(vrt, d, "scene.obj")        # textured blob (uv outside [0,1]) + plain floor, v/vn/vt per corner, two materials
    ours = vrt.scene.load_obj(os.path.join(d, "scene.obj"))
    ref = po.ref_scene([os.path.join(d, "scene.obj")])

    def canon(sc):
        tri = np.ascontiguousarray(sc["tri"]).view(np.uint8).reshape(-1, 36)
        ex = np.ascontiguousarray(sc["triEx"]).view(np.uint8).reshape(-1, 64)
        rec = np.concatenate([tri, ex], 1)
        return rec[np.lexsort(rec.T[::-1])]
    a, b = canon(ours), canon(ref)
    assert a.shape == b.shape and len(a) > 1000
    assert np.array_equal(a, b)
    assert np.array_equal(np.asarray(ours["tex"]), np.asarray(ref["tex"])) and len(ref["tex"]) == 23 * 37 * 4
    mat_dt = np.dtype([("f", "<f4", 16), ("tex_id", "<i4"), ("illum", "<i4"), ("tw", "<u4"), ("th", "<u4"), ("off", "<u8")])
    ma, mb = np.ascontiguousarray(ours["mat"]).view(mat_dt), np.ascontiguousarray(ref["mat"]).view(mat_dt)
    assert len(ma) == len(mb) == 2
    for k in ("f", "tex_id", "tw", "th"):
        assert np.array_equal(ma[k], mb[k]), k
    textured = mb["tex_id"] >= 0
    assert textured.any() and np.array_equal(ma["off"][textured], mb["off"][textured])


def test_rng_equals_the_reference_helpers_live(po):
    """Fresh seeds (not the fixture's): orc_rng == vxref_rng (the reference's WangHash / RandomInt / RandomFloat, common.h:129-147)."""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rs = np.random.RandomState(99)
    for seed in [int(x) for x in rs.randint(0, 2 ** 32, size=40, dtype=np.uint64)] + [0, 0xFFFFFFFF]:
        a, b = po.rng(seed, 512), po.rng(seed, 512, po.ref())
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_mirror_fixtures_equal_the_reference_twin_live(po, golden, tmp_path):
    """tests/golden/mirror_*.npz regenerated from the reference's object code now (oracle/gen_golden_mirror.py: both reference scene
    builders on the same OBJs, the twin's Trace on the RTU camera rays): same buffers, same radiance, bit for bit."""
    if not po.have_ref_rc():
        pytest.skip("oracle/_ref/libvxref_rc.so not built")
    import importlib
    gm = importlib.import_module("oracle.gen_golden_mirror")
    gm.OUT = str(tmp_path)
    gm.TMP = str(tmp_path / "objs")
    gm.main()
    for name in ("mirror_teapot", "mirror_trio"):
        want = golden(name)
        with np.load(tmp_path / (name + ".npz")) as z:
            assert sorted(z.files) == sorted(want.keys())
            for k in z.files:
                assert np.atleast_1d(z[k]).tobytes() == np.atleast_1d(want[k]).tobytes(), (name, k)
