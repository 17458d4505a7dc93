"""Regenerate tests/golden/*.npz from the REFERENCE's own object code (oracle/_ref/libvxref.so).

TEST INFRASTRUCTURE.  Needs /root/reference (assets + sources); run in the build container only:
    python -m oracle.gen_golden
Each fixture holds data only: the byte images of the buffers the reference scene builder produced
for one of the reference's own test assets (tests/regression/raytracing/assets/*.obj), a seeded ray
set, and what the reference BVHTraverser / shading helpers returned for it:
    hits      closest hit per ray        (rt_traversal.cpp + accept loop, bit pattern)
    anyhits   first accepted candidate   (same traverser, terminated after the first ACCEPT)
    colors    f32 radiance per ray       (closest.cpp:57-127 through rtx_shading.h helpers)
    rgb8      packed pixel               (common.h:149-154)
Ray counts are small on the axis-aligned assets because the reference traverser spins for 2^32
iterations on some of their rays."""
import os
import sys

import numpy as np

from . import pyoracle as po

ASSETS = "/root/reference/tests/regression/raytracing/assets/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TMP = "/tmp/vx_golden_objs"


def wrap(name):
    """The shipped assets have no MTL, and the RTU closest-hit shader dereferences mat[texId]
    unconditionally (closest.cpp:55): give them one material so shading is defined."""
    os.makedirs(TMP, exist_ok=True)
    with open(os.path.join(TMP, "m.mtl"), "w") as f:
        f.write("newmtl m0\nKd 0.7 0.6 0.5\nKa 0.1 0.1 0.1\n")
    p = os.path.join(TMP, name)
    with open(p, "w") as f:
        f.write("mtllib m.mtl\nusemtl m0\n" + open(ASSETS + name).read())
    return p


def wrap_textured(name, png, uv_scale, uv_shift, tag):
    """The asset with an MTL whose diffuse map is one of the reference's own PNGs, and its texture coordinates stretched and
    shifted so that they leave [0,1] on both sides (negative and > 1): texSample's `uint32_t(uv * w) % w` (rtx_shading.h:5-18)
    and RGB8toRGB32F (common.h:156-162) are then exercised as the reference's object code evaluates them."""
    import shutil
    os.makedirs(TMP, exist_ok=True)
    shutil.copy(ASSETS + png, os.path.join(TMP, png))
    mtl = "m_%s.mtl" % tag
    with open(os.path.join(TMP, mtl), "w") as f:
        f.write("newmtl mt_%s\nKd 0.9 0.8 0.7\nKa 0.1 0.1 0.1\nmap_Kd %s\n" % (tag, png))
    out = []
    for line in open(ASSETS + name):
        if line.startswith("vt "):
            u, v = [float(x) for x in line.split()[1:3]]
            line = "vt %.6f %.6f\n" % (u * uv_scale + uv_shift, v * uv_scale - uv_shift * 0.5)
        out.append(line)
    p = os.path.join(TMP, tag + "_" + name)
    with open(p, "w") as f:
        f.write("mtllib %s\nusemtl mt_%s\n" % (mtl, tag) + "".join(out))
    return p


def rays_at(scene, n, seed):
    tri = scene["tri"].view(np.float32).reshape(-1, 3)
    blas = scene["blas"].view(np.float32).reshape(-1, 40)
    lo, hi = tri.min(0), tri.max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    nb = len(blas)
    if nb > 1:   # instances are translated (scene.cpp:214-250): aim at all of them
        tr = blas[:, 17:33].reshape(nb, 4, 4)[:, :3, 3]
        c = c + tr.mean(0)
        r = r + np.linalg.norm(tr - tr.mean(0), axis=1).max()
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = c + o / np.linalg.norm(o, axis=1, keepdims=True) * r * 2.0
    tgt = c + rng.uniform(-1, 1, size=(n, 3)) * r * 0.7
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    # a few axis-parallel rays: zero direction components exercise the NaN paths of the slab test
    k = max(4, n // 16)
    ax = np.zeros((k, 6), np.float32)
    ax[:, :3] = (c + np.array([-2 * r, 0, 0]))[None] + rng.uniform(-0.5, 0.5, size=(k, 3)) * r * np.array([0, 1, 1])
    ax[:, 3] = 1.0
    return np.concatenate([rays, ax])


def edge_rays(scene):
    """Axis-parallel rays through the midpoints of triangle edges of an axis-aligned mesh: adjacent
    triangles then report bit-identical distances, so the winner is decided by traversal order."""
    tri = scene["tri"].view(np.float32).reshape(-1, 3, 3)
    out = []
    for t in tri:
        n = np.cross(t[1] - t[0], t[2] - t[0])
        a = int(np.argmax(np.abs(n)))
        sgn = 1.0 if n[a] > 0 else -1.0
        for i in range(3):
            m = (t[i] + t[(i + 1) % 3]) * 0.5
            o = m.copy()
            o[a] += 3.0 * sgn
            d = np.zeros(3, np.float32)
            d[a] = -sgn
            out.append(np.concatenate([o, d]))
    return np.unique(np.array(out, np.float32), axis=0)


def make(name, objs, n, seed, edges=False):
    sc = po.ref_scene(objs)
    rays = rays_at(sc, n, seed)
    if edges:
        rays = np.concatenate([rays, edge_rays(sc)])
    hits, st = po.trace_ref(sc, rays)
    anyh, _ = po.trace_ref(sc, rays, any_hit=True)
    col, px = po.ref_shade(sc, rays, hits)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), rays=rays, hits=hits, anyhits=anyh, colors=col, rgb8=px,
                        ref_node_reads=np.uint64(st["node_reads"]), ref_tri_reads=np.uint64(st["tri_reads"]), **sc)
    print(name, "rays", len(rays), "hits", int((hits["dist"] < 1e29).sum()), st, flush=True)


def camera_fixture():
    L = po.ref()
    out = []
    for (w, h) in ((64, 48), (40, 32), (1920, 1080)):
        xs = range(w) if w <= 64 else (0, 1, 959, 960, 961, 1919)
        ys = range(h) if h <= 64 else (0, 1, 539, 540, 541, 1079)
        for y in ys:
            for x in xs:
                r = np.zeros(6, np.float32)
                L.vxref_generate_ray(x, y, w, h, r.ctypes.data)
                out.append((x, y, w, h) + tuple(r.view(np.uint32)))
    np.savez_compressed(os.path.join(OUT, "camera_rays.npz"), rows=np.array(out, np.uint32))
    print("camera_rays", len(out), flush=True)


RNG_SEEDS = (0, 1, 61, 12345, 0x9E3779B9, 0xFFFFFFFF, 2073600 * 16 + 7)


def rng_fixture():
    """The reference's WangHash / RandomInt / RandomFloat (common.h:129-147) through oracle/_ref: 256 values per seed."""
    L = po.ref()
    out = {}
    for k, seed in enumerate(RNG_SEEDS):
        h, i, f = po.rng(seed, 256, L)
        out["hash%d" % k], out["ints%d" % k], out["floats%d" % k] = h, i, f
    np.savez_compressed(os.path.join(OUT, "rng.npz"), seeds=np.array(RNG_SEEDS, np.uint64), **out)
    print("rng", len(RNG_SEEDS), "seeds", flush=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "cube":
        make("cube", [wrap("cube.obj")], 32, 16, edges=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "textured":
        make_textured()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rng":
        rng_fixture()
        return
    camera_fixture()
    rng_fixture()
    make("teapot", [wrap("teapot.obj")], 4096, 11)
    make("torus", [wrap("torus.obj")], 2048, 12)
    make("sphere", [wrap("sphere.obj")], 2048, 13)
    make("teapot_x3", [wrap("teapot.obj")] * 3, 2048, 17)   # TLAS with internal nodes + translated instances
    make("sphere_x6", [wrap("sphere.obj")] * 6, 1024, 18)   # TLAS deeper than one level
    make("cube", [wrap("cube.obj")], 32, 16, edges=True)
    make("cone", [wrap("cone.obj")], 96, 14)
    make("cylinder", [wrap("cylinder.obj")], 96, 15)
    make_textured()


def make_textured():
    # textured materials (two PNGs of different, non-power-of-two sizes) next to an untextured one, uv outside [0,1]:
    # pins texSample / RGB8toRGB32F and the texId rebasing of scene.cpp through the reference's own code
    make("tex_mix", [wrap_textured("torus.obj", "flower.png", 3.7, -1.3, "a"), wrap("sphere.obj"),
                     wrap_textured("teapot.obj", "red.png", 2.2, -0.6, "b")], 3072, 21)


# ---------------------------------------------------------------------------------------------
# software twin (tests/regression/raycast): fixtures from the reference's own CPU path
# ---------------------------------------------------------------------------------------------
RC_ASSETS = "/root/reference/tests/regression/raycast/assets/"


def gen_raycast():
    """tests/golden/rc_*.npz: buffers built by the reference's raycast Scene/BVH/TLAS code, the camera of
    Tracer::setup, and what the reference's own GenerateRay/Trace (its -c path) rendered; plus hit records of
    TLASIntersect for a subset of the camera rays.  red.png travels as a data file with the texels stb_image
    decoded from it (surface.cpp:28-55), to pin the package's own PNG decoder."""
    cases = [
        ("rc_teapot", ["teapot.obj"], ["red.png"], [0.0], 80, 60, 1, 1),
        ("rc_teapot_x3", ["teapot.obj"] * 3, ["green.png", "red.png", "flower.png"], [0.0, 0.5, 0.3], 96, 64, 2, 3),
        ("rc_torus_x2", ["torus.obj"] * 2, ["green.png", "blue.png"], [0.6, 0.4], 64, 48, 1, 4),
        ("rc_cube_x2", ["cube.obj"] * 2, ["blue.png", "red.png"], [0.2, 0.9], 64, 48, 3, 5),
    ]
    for name, objs, texs, refl, w, h, spp, depth in cases:
        sc = po.RefRcScene([RC_ASSETS + o for o in objs], [RC_ASSETS + t for t in texs], refl)
        cam = sc.camera(45.0, 1.0, w, h)
        light = np.array(po.RC_DEFAULT_LIGHT, np.float32)
        px = sc.render(w, h, spp, depth, cam, light)
        b = dict(sc.buffers)
        b["tlas_root"] = sc.tlas_root
        args = po.rc_args(b, w, h, cam, light, spp, depth)
        rays = po.rc_camera_rays(args)[::5]
        hits = sc.trace(rays)
        out = dict(sc.buffers)
        out.update(tlas_root=np.uint32(sc.tlas_root), cam14=cam, light12=light, width=np.uint32(w), height=np.uint32(h),
                   spp=np.uint32(spp), max_depth=np.uint32(depth), pixels=px, rays=rays, hits=hits)
        if name == "rc_teapot":
            out["red_png"] = np.frombuffer(open(RC_ASSETS + "red.png", "rb").read(), np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k in ("tri", "bvh", "tex", "pixels", "hits")},
              "non-background %.3f" % (px != px[0, 0]).mean(), "hits", int((hits["dist"] < 1e29).sum()))
        sc.close()


if __name__ == "__main__":
    if "--raycast" in sys.argv:      # python -m oracle.gen_golden --raycast : only the software-twin fixtures
        gen_raycast()
    else:
        main()
        gen_raycast()
