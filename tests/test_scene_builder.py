"""CPU: the host-side BVH4 builder / quantiser (csrc/scene_builder.cpp).  Its tree shape is its own
(the reference widens clusters from uninitialised bounds, bvh.cpp:79-86), so it is validated by
invariants and by traversal equivalence against brute force and against the reference-built tree."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NODE = np.dtype([("o", "<f4", 3), ("e", "i1", 3), ("imask", "u1"), ("lf", "<u4"), ("ld", "<u4"), ("ch", "u1", (4, 7))])
assert NODE.itemsize == 52


def decode(n, k):
    q = n["ch"][k, 1:].astype(np.float32)
    s = np.ldexp(np.float32(1), n["e"].astype(np.int32)).astype(np.float32)
    return n["o"] + q[:3] * s, n["o"] + q[3:] * s


def check_tree(sc):
    nodes = sc["bvh"].view(NODE)
    tri = sc["tri"].view(np.float32).reshape(-1, 3, 3)
    seen = np.zeros(len(tri), bool)
    stack = [(0, None, None, 0)]
    maxd = 0
    while stack:
        i, lo, hi, d = stack.pop()
        maxd = max(maxd, d)
        n = nodes[i]
        assert n["imask"] == 0
        if n["ld"] != 0:      # leaf
            t = tri[n["lf"]: n["lf"] + n["ld"]]
            assert not seen[n["lf"]: n["lf"] + n["ld"]].any()
            seen[n["lf"]: n["lf"] + n["ld"]] = True
            if lo is not None:   # conservative: every vertex inside the decoded box chain
                assert (t >= lo - 0).all() and (t <= hi + 0).all()
        else:
            kids = [k for k in range(4) if n["ch"][k, 0] != 0]
            assert len(kids) >= 2 and kids == list(range(len(kids)))
            for k in kids:
                clo, chi = decode(n, k)
                assert (chi >= clo).all()
                stack.append((int(n["lf"]) + k, clo if lo is None else np.maximum(clo, lo), chi if hi is None else np.minimum(chi, hi), d + 1))
    assert seen.all()
    return maxd


def check_tree_fast(sc, root=0):
    """check_tree for trees of millions of nodes: the same properties, breadth-first with numpy over whole levels -- every
    triangle in exactly one leaf, every vertex inside the decoded boxes of ALL its ancestors (the intersection of the chain),
    internal nodes with >= 2 children in the first slots, children after their parent.  Returns the depth."""
    nodes = sc["bvh"].view(NODE)
    tri = sc["tri"].view(np.float32).reshape(-1, 3, 3)
    seen = np.zeros(len(tri), np.int32)
    idx = np.array([root], np.int64)
    lo = np.full((1, 3), -np.inf, np.float32)
    hi = np.full((1, 3), np.inf, np.float32)
    depth = 0
    while len(idx):
        n = nodes[idx]
        assert (n["imask"] == 0).all()
        leaf = n["ld"] != 0
        if leaf.any():
            lf, ld = n["lf"][leaf].astype(np.int64), n["ld"][leaf].astype(np.int64)
            assert (lf + ld <= len(tri)).all()
            owner = np.repeat(np.arange(len(lf)), ld)                       # leaf of every listed triangle
            t = np.repeat(lf - np.concatenate([[0], np.cumsum(ld)[:-1]]), ld) + np.arange(int(ld.sum()))
            np.add.at(seen, t, 1)
            llo, lhi = lo[leaf][owner], hi[leaf][owner]
            v = tri[t]                                                       # [k, 3 vertices, 3]
            assert (v >= llo[:, None, :]).all() and (v <= lhi[:, None, :]).all(), "a vertex outside a decoded box of its ancestors"
        inner = ~leaf
        if not inner.any():
            break
        ni, nn = idx[inner], n[inner]
        valid = nn["ch"][:, :, 0] != 0                                       # [m, 4]
        cnt = valid.sum(1)
        assert (cnt >= 2).all() and (valid == (np.arange(4)[None, :] < cnt[:, None])).all(), "children not in the first slots"
        s = np.ldexp(np.float32(1), nn["e"].astype(np.int32)).astype(np.float32)          # [m, 3]
        q = nn["ch"][:, :, 1:].astype(np.float32)                                           # [m, 4, 6]
        clo = nn["o"][:, None, :] + q[:, :, :3] * s[:, None, :]
        chi = nn["o"][:, None, :] + q[:, :, 3:] * s[:, None, :]
        assert (chi >= clo)[valid].all()
        clo = np.maximum(clo, lo[inner][:, None, :])
        chi = np.minimum(chi, hi[inner][:, None, :])
        child = nn["lf"].astype(np.int64)[:, None] + np.arange(4)[None, :]
        assert (child[valid] > np.repeat(ni, cnt)).all() and (child[valid] < len(nodes)).all(), "children must come after their parent"
        idx, lo, hi = child[valid], clo[valid], chi[valid]
        depth += 1
    assert (seen == 1).all(), "triangles in no leaf: %d, in several: %d" % (int((seen == 0).sum()), int((seen > 1).sum()))
    return depth


@pytest.mark.parametrize("args", [("cornell", 0, 0, 1), ("blob", 3, 0, 2), ("atrium", 4, 0, 3), ("hairball", 60, 20, 7)])
def test_builder_invariants(vrt, args):
    sc = vrt.scene.procedural(*args)
    d = check_tree(sc)
    assert d == sc.info["max_depth"] and d < 32        # the reference's trail supports 32 levels
    assert check_tree_fast(sc) == d
    assert sc.info["n_tris"] == sc.n_tris
    tl = sc["tlas"].view(NODE)
    assert len(tl) == 1 and tl[0]["imask"] == 1 and tl[0]["ld"] == 0     # single mesh: TLAS root is the instance leaf (bvh.cpp:325-328)


def test_every_builder_variant_keeps_the_invariants():
    """The builder's stages one by one (the knobs are read when the library loads, so child processes): greedy widening while building
    (VXS_COLLAPSE=0, the builder up to round 3's r03_s), binary SAH tree + the SAH dynamic programme's 4-wide collapse, the default
    (+ two passes of reinsertion), and the subtree-parallel reinsertion: same invariants, same closest distances."""
    import subprocess, sys
    code = (
        "import sys, importlib, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "vrt = importlib.import_module('vortex-raytracing_amd')\n"
        "from test_scene_builder import check_tree, check_tree_fast\n"
        "from oracle import pyoracle as po\n"
        "for args in (('atrium', 6, 0, 3), ('hairball', 60, 20, 7), ('blob', 3, 0, 2)):\n"
        "    sc = vrt.scene.procedural(*args)\n"
        "    d = check_tree(sc)\n"
        "    assert d == sc.info['max_depth'] < 32 and check_tree_fast(sc) == d and sc.info['max_leaf'] <= 4\n"
        "    rays = po.camera_rays(64, 48)\n"
        "    rays = rays[(rays[:, 3:] != 0).all(1)]\n"
        "    h, _ = po.trace_canonical(sc, rays)\n"
        "    np.save(sys.argv[1] + '_' + args[0] + '.npy', h['dist'])\n"
        "    print(args[0], sc.n_bvh_nodes)\n") % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = {}
        for mode, kv in (("0", {"VXS_COLLAPSE": "0"}), ("1", {"VXS_COLLAPSE": "1", "VXS_OPTIMIZE": "0"}), ("2", {}), ("3", {"VXS_OPTIMIZE_LOCAL": "0"})):
            env = dict(os.environ, **kv)
            r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "m" + mode)], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[mode] = dict(l.split() for l in r.stdout.strip().splitlines())
        hits = 0
        for name in ("atrium", "hairball", "blob"):
            a = np.load(os.path.join(d, "m0_%s.npy" % name))
            hits += int((a < 1e29).sum())
            for m in ("1", "2", "3"):
                b = np.load(os.path.join(d, "m%s_%s.npy" % (m, name)))
                assert (a == b).mean() > 0.999      # (a re-quantised box chain can drop a grazing hit in either tree: DESIGN.md s3)
        assert hits > 300
        assert len({tuple(sorted(o.items())) for o in outs.values()}) == 4          # every variant really built a different tree


def test_atrium_level8_is_the_1m_triangle_scene(vrt):
    sc = vrt.scene.procedural("atrium", 8, 0, 3)
    assert sc.n_tris == 1048576
    assert sc.info["max_depth"] < 32
    assert check_tree_fast(sc) == sc.info["max_depth"]      # the exact containment check at the size the benchmark runs
    assert sc.n_mats == 16 and sc["tex"].size == 8 * 256 * 256 * 4


def brute_force(sc, rays, po):
    import ctypes as C
    L = po.orc()
    L.orc_ray_tri.restype = C.c_float
    L.orc_ray_tri.argtypes = [C.c_void_p] * 5
    tri = sc["tri"].view(np.float32).reshape(-1, 9)
    out = np.full(len(rays), np.float32(1e30))
    b = (C.c_float * 3)()
    for i, r in enumerate(rays):
        for t in tri:
            d = np.float32(L.orc_ray_tri(r.ctypes.data, t.ctypes.data, C.byref(b, 0), C.byref(b, 4), C.byref(b, 8)))
            if d < out[i]:
                out[i] = d
    return out


def test_traversal_of_our_tree_finds_the_brute_force_distance(vrt, po):
    sc = vrt.scene.procedural("blob", 2, 0, 5)        # 320 triangles
    rays = po.camera_rays(40, 30)
    # a zero direction component makes 0*inf = NaN slabs; the reference's std::min/std::max then reject
    # boxes the ray grazes, so "closest" is only defined for rays without zero components
    rays = rays[(rays[:, 3:] != 0).all(1)]
    hits, st = po.trace_faithful(sc, rays)
    want = brute_force(sc, rays, po)
    assert (hits["dist"] < 1e29).sum() > 50
    assert np.array_equal(hits["dist"], want)
    c, _ = po.trace_canonical(sc, rays)
    assert np.array_equal(c.view(np.uint8), hits.view(np.uint8))


def test_same_triangles_reference_tree_and_our_tree_agree_on_distance(vrt, po, golden):
    """Reference-built teapot vs our builder on the same triangle soup: same closest distances
    (indices differ because both builders reorder triangles)."""
    g = golden("teapot")
    tri = g["tri"].view(np.float32).reshape(-1, 9)
    ours = vrt.scene.from_triangles([tri])
    a, _ = po.trace_canonical(ours, g["rays"])
    assert np.array_equal(a["dist"], g["hits"]["dist"])
    # and the hit triangle is the same geometry
    ta = ours["tri"].view(np.float32).reshape(-1, 9)[a["triIdx"]]
    tb = tri[g["hits"]["triIdx"]]
    hit = g["hits"]["dist"] < 1e29
    assert np.array_equal(ta[hit], tb[hit])


def test_multi_instance_scene_from_triangles(vrt, po):
    rng = np.random.default_rng(3)
    base = rng.uniform(-1, 1, size=(64, 9)).astype(np.float32)
    xf = []
    for i in range(6):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = (220 + 30 * i, 100 + 25 * ((i % 3) - 1), -150 + 60 * i)
        m[:3, :3] *= 35.0
        xf.append(m)
    sc = vrt.scene.from_triangles([base] * 6, xf)
    assert sc.n_blas == 6 and sc.n_tlas_nodes > 6
    rays = po.camera_rays(64, 48)
    c, st = po.trace_canonical(sc, rays)
    assert (c["dist"] < 1e29).sum() > 20 and len(set(c["blasIdx"][c["dist"] < 1e29])) >= 4
    # brute force in world space: transform triangles by each instance matrix
    tri = np.concatenate([(base.reshape(-1, 3) @ m[:3, :3].T + m[:3, 3]).reshape(-1, 9) for m in xf]).astype(np.float32)
    world = vrt.scene.from_triangles([tri])
    w, _ = po.trace_canonical(world, rays)
    assert np.array_equal(c["dist"] < 1e29, w["dist"] < 1e29)
    np.testing.assert_allclose(c["dist"], w["dist"], rtol=2e-5)


# ---- image / texture ingest (surface.cpp:28-55 packs (r << 16) + (g << 8) + b from 3 forced channels) ----
def _png_bytes(rows, w, h, depth, ctype, interlace=False, palette=None):
    """Minimal PNG writer for the tests: rows = list of h raw scanlines (bytes, already packed for depth/ctype).
    Cycles through the five filter types so that every unfilter path is exercised."""
    import struct, zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = ch * depth
    bpp = max(1, bits // 8)

    def filt(rows_):
        out, prev = bytearray(), None
        for y, row in enumerate(rows_):
            f = y % 5
            prev = prev if prev is not None else bytes(len(row))
            enc = bytearray()
            for i, v in enumerate(row):
                a = row[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 0: p = 0
                elif f == 1: p = a
                elif f == 2: p = b
                elif f == 3: p = (a + b) >> 1
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                enc.append((v - p) & 255)
            out.append(f); out += enc
            prev = row
        return bytes(out)

    def sub_rows(x0, y0, dx, dy):
        """extract the sub-image of one Adam7 pass from unpacked pixel tuples"""
        res = []
        for y in range(y0, h, dy):
            px = [pixels[y][x] for x in range(x0, w, dx)]
            if px:
                res.append(pack(px))
        return res

    def unpack(row):
        if depth >= 8:
            n = depth // 8 * ch
            return [bytes(row[i * n:(i + 1) * n]) for i in range(w)]
        per = 8 // depth
        return [(row[x // per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1) for x in range(w)]

    def pack(px):
        if depth >= 8:
            return b"".join(px)
        per, out = 8 // depth, bytearray((len(px) * depth + 7) // 8)
        for x, v in enumerate(px):
            out[x // per] |= v << ((per - 1 - x % per) * depth)
        return bytes(out)

    if interlace:
        pixels = [unpack(r) for r in rows]
        raw = b"".join(filt(sub_rows(*ps)) for ps in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
                       if sub_rows(*ps))
    else:
        raw = filt(rows)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    z = zlib.compress(raw, 6)
    out += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    return out


@pytest.mark.parametrize("interlace", [False, True])
def test_png_decoder_all_colour_types_filters_and_adam7(vrt, tmp_path, interlace):
    rng = np.random.default_rng(5)
    w, h = 13, 11   # not multiples of 8: partial Adam7 passes and partial bytes at low bit depths
    cases = []
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    cases.append(("rgb8", 8, 2, [bytes(r.tobytes()) for r in rgb], None, rgb))
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    cases.append(("rgba8", 8, 6, [bytes(r.tobytes()) for r in rgba], None, rgba[..., :3]))
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    cases.append(("grey8", 8, 0, [bytes(r.tobytes()) for r in g], None, np.repeat(g[..., None], 3, 2)))
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    cases.append(("greya8", 8, 4, [bytes(r.tobytes()) for r in ga], None, np.repeat(ga[..., :1], 3, 2)))
    rgb16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    cases.append(("rgb16", 16, 2, [bytes(r.astype(">u2").tobytes()) for r in rgb16], None, (rgb16 >> 8).astype(np.uint8)))
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    def pack4(r):
        out = bytearray((w * 4 + 7) // 8)
        for x, v in enumerate(r):
            out[x // 2] |= int(v) << (4 if x % 2 == 0 else 0)
        return bytes(out)
    cases.append(("pal4", 4, 3, [pack4(r) for r in idx], pal.reshape(-1).tolist(), pal[idx]))
    g1 = rng.integers(0, 2, (h, w), dtype=np.uint8)
    def pack1(r):
        out = bytearray((w + 7) // 8)
        for x, v in enumerate(r):
            out[x // 8] |= int(v) << (7 - x % 8)
        return bytes(out)
    cases.append(("grey1", 1, 0, [pack1(r) for r in g1], None, np.repeat((g1 * 255)[..., None], 3, 2)))
    for name, depth, ctype, rows, palette, want in cases:
        f = tmp_path / (name + ".png")
        f.write_bytes(_png_bytes(rows, w, h, depth, ctype, interlace, palette))
        got = vrt.scene.image_load(f)
        exp = (want[..., 0].astype(np.uint32) << 16) + (want[..., 1].astype(np.uint32) << 8) + want[..., 2].astype(np.uint32)
        np.testing.assert_array_equal(got, exp, err_msg=name)
        try:
            from PIL import Image
        except ImportError:
            continue
        ref = np.asarray(Image.open(f).convert("RGB")) if depth != 16 else None   # our own writer agrees with an independent reader
        if ref is not None:
            np.testing.assert_array_equal(ref, want, err_msg=name + " (PIL)")
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    with pytest.raises(ValueError):
        vrt.scene.image_load(bad)


def test_pnm_reader(vrt, tmp_path):
    px = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3) * 9
    (tmp_path / "a.ppm").write_bytes(b"P6\n# comment\n3 2\n255\n" + px.tobytes())
    (tmp_path / "b.ppm").write_text("P3\n3 2\n255\n" + " ".join(str(int(v)) for v in px.reshape(-1)) + "\n")
    (tmp_path / "c.pgm").write_bytes(b"P5 3 2 255\n" + px[..., 0].tobytes())
    exp = (px[..., 0].astype(np.uint32) << 16) + (px[..., 1].astype(np.uint32) << 8) + px[..., 2]
    np.testing.assert_array_equal(vrt.scene.image_load(tmp_path / "a.ppm"), exp)
    np.testing.assert_array_equal(vrt.scene.image_load(tmp_path / "b.ppm"), exp)
    gexp = px[..., 0].astype(np.uint32) * 0x010101
    np.testing.assert_array_equal(vrt.scene.image_load(tmp_path / "c.pgm"), gexp)


def test_obj_mtl_with_map_kd_feeds_the_texture_buffer(vrt, po, tmp_path):
    """mesh.cpp:130-293 / scene.cpp:61-80: a material with map_Kd points at its texels in the shared
    texture buffer; the oracle's texSample then returns those texels."""
    rng = np.random.default_rng(9)
    tex = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    (tmp_path / "t.png").write_bytes(_png_bytes([bytes(r.tobytes()) for r in tex], 8, 8, 8, 2))
    (tmp_path / "m.mtl").write_text("newmtl plain\nKd 0.2 0.4 0.6\nnewmtl tex\nKd 1 1 1\nmap_Kd -s 1 1 1 t.png\nnewmtl missing\nmap_Kd nope.png\n")
    (tmp_path / "q.obj").write_text(
        "mtllib m.mtl\nv 300 0 -100\nv 300 0 100\nv 300 200 100\nv 300 200 -100\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn -1 0 0\n"
        "usemtl tex\nf 1/1/1 2/2/1 3/3/1 4/4/1\nusemtl plain\nv 300 -50 -100\nv 300 -50 100\nf 5//1 6//1 2//1 1//1\n"
        "usemtl missing\nv 300 250 -100\nv 300 250 100\nf 4//1 3//1 8//1 7//1\n")
    sc = vrt.scene.load_obj(tmp_path / "q.obj")
    mats = np.frombuffer(bytes(sc.buffers["mat"]), np.uint8).reshape(-1, 88)
    tex_id = mats[:, 64:68].copy().view(np.int32).reshape(-1)
    dims = mats[:, 72:80].copy().view(np.uint32).reshape(-1, 2)
    offs = mats[:, 80:88].copy().view(np.uint64).reshape(-1)
    assert list(tex_id) == [-1, 0, -1] and tuple(dims[1]) == (8, 8) and offs[1] == 0
    texels = np.frombuffer(bytes(sc.buffers["tex"]), np.uint32)
    exp = (tex[..., 0].astype(np.uint32) << 16) + (tex[..., 1].astype(np.uint32) << 8) + tex[..., 2]
    np.testing.assert_array_equal(texels[:64].reshape(8, 8), exp)
    # a frame through the oracle shows texel colours on the textured quad and the flat Kd below it
    w, h = 64, 48
    px, hits, col = po.render(sc, w, h, po.shade_params(light_pos=(0.0, 100.0, 0.0)))
    assert (hits["dist"] < 1e29).mean() > 0.1
    lit = col[hits["dist"] < 1e29]
    assert len(np.unique(lit.round(4), axis=0)) > 10      # many distinct texel colours, not one flat material


@pytest.mark.parametrize("name", ["teapot", "torus", "sphere", "cone", "cylinder"])
def test_tree_quality_anchored_to_the_reference_builder(vrt, po, golden, name):
    """The committed fixtures hold the tree the REFERENCE's builder made of its own assets (bvh.cpp:30-264 through oracle/_ref).  On the
    same triangles and the fixture's rays, the package's SAH tree must not cost more algorithmic bytes per ray (52 B per node fetch,
    36 B per triangle test) than the reference's: it spends more triangle tests (leaves of up to 4) to save node
    fetches.  tests/tree_quality.py prints the same comparison at the benchmark's scale (profiles/r03_d_tree_quality.txt)."""
    g = golden(name)
    tri = g["tri"].view(np.float32).reshape(-1, 9)
    ours = vrt.scene.from_triangles([tri])
    a, sa = po.trace_canonical(g, g["rays"])
    b, sb = po.trace_canonical(ours, g["rays"])
    assert np.array_equal(a["dist"], b["dist"])
    ref_bytes = 52 * sa["node_reads"] + 36 * sa["tri_reads"]
    our_bytes = 52 * sb["node_reads"] + 36 * sb["tri_reads"]
    print("%s: reference tree %.1f nodes + %.2f tris per ray = %d B; ours %.1f + %.2f = %d B" % (
        name, sa["node_reads"] / len(a), sa["tri_reads"] / len(a), ref_bytes // len(a), sb["node_reads"] / len(b), sb["tri_reads"] / len(b), our_bytes // len(b)))
    assert our_bytes <= 1.02 * ref_bytes
    assert sb["node_reads"] < sa["node_reads"]


def test_builder_on_small_and_degenerate_meshes(vrt, po):
    """Meshes of 1..257 random triangles, 300 coincident ones (no split separates them), a flat axis-aligned sheet, six triangles with
    one centroid: the tree keeps the invariants and every ray finds the brute-force distance (the binary tree + reinsertion + collapse
    path has to cope with leaves that cannot be split and with subtrees of two or three nodes)."""
    rng = np.random.default_rng(7)

    def soup(n, spread=100.0, size=10.0):
        c = rng.uniform(-spread, spread, size=(n, 1, 3)).astype(np.float32) + np.array([200, 100, 0], np.float32)
        return (c + rng.uniform(-size, size, size=(n, 3, 3)).astype(np.float32)).reshape(n, 9)

    cases = [("soup%d" % n, soup(n)) for n in (1, 2, 3, 4, 5, 7, 9, 17, 64, 257)]
    cases.append(("coincident", np.tile(soup(3), (100, 1))))
    g = np.linspace(-50, 50, 11, dtype=np.float32)
    sheet = []
    for i in range(10):
        for j in range(10):
            a, b, c, d = ([g[j] + 200, 100, g[i]], [g[j + 1] + 200, 100, g[i]], [g[j] + 200, 100, g[i + 1]], [g[j + 1] + 200, 100, g[i + 1]])
            sheet += [a + b + c, b + d + c]
    cases.append(("flat sheet", np.array(sheet, np.float32)))
    cases.append(("one centroid", np.tile(np.array([[190, 90, -10, 210, 90, -10, 200, 120, 10]], np.float32), (6, 1))))
    rays = po.camera_rays(24, 18)
    rays = rays[(rays[:, 3:] != 0).all(1)]
    hits = 0
    for name, tri in cases:
        sc = vrt.scene.from_triangles([tri])
        d = check_tree(sc)
        assert d == sc.info["max_depth"] < 32 and check_tree_fast(sc) == d, name
        h, _ = po.trace_canonical(sc, rays)
        assert np.array_equal(h["dist"], brute_force(sc, rays, po)), name
        hits += int((h["dist"] < 1e29).sum())
    assert hits > 20


@pytest.mark.parametrize("name,args,threads", [("atrium", (7, 0, 3), (1, 3, 8)), ("hairball", (4200, 250, 7), (2, 8))])
def test_the_built_scene_does_not_depend_on_the_thread_count(vrt, monkeypatch, name, args, threads):
    """Every parallel piece of the builder (nodes of >= 131,072 triangles shared by the threads, subtrees built and optimised one thread
    each, the passes over the finished tree by subtrees) writes disjoint data per job and reduces in job order: the bytes of every buffer
    are the same at any thread count.  262,144 triangles take the schedule for scenes up to 2 M triangles, 2.1 M the one above."""
    monkeypatch.delenv("VXRT_SCENE_CACHE", raising=False)
    ref = None
    for t in threads:
        monkeypatch.setenv("VXS_THREADS", str(t))
        sc = vrt.scene.procedural(name, *args)
        bufs = {k: np.ascontiguousarray(sc[k]).view(np.uint8).tobytes() for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat")}
        if ref is None:
            ref = bufs
            assert sc.n_tris >= 262144
        else:
            for k in bufs:
                assert bufs[k] == ref[k], (k, t)
