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
