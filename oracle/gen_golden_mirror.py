"""tests/golden/mirror_*.npz: the pin of the RTU path's MIRROR ARM (shaders/closest.cpp:95-121) to reference object code.

TEST INFRASTRUCTURE.  Needs /root/reference; run in the build container only:   python -m oracle.gen_golden_mirror

closest.cpp only builds for RISC-V, and the RTU test's own scene builder hard-codes reflectivity 0 (raytracing/scene.cpp:96), so no
reference run of that shader with a bounce can exist here.  The same formulas -- R = normalize(d - 2 N (N.d)), origin I + R * 0.001,
radiance = diffuse (1 - r) + r * next, background when the depth is used up -- are host code in the reference's software twin
(raycast/render.h:210-277, reflectivity per instance as raycast/tracer.cpp:13: 0.5 / 0.3), compiled where it lies into
oracle/_ref/libvxref_rc.so.  One scene is put through BOTH reference builders:
  * meshes = the reference's own assets, written out again scaled and moved in front of the RTU kernel's fixed camera (eye (0,100,0)
    looking along +x, raytracing/kernel.cpp:28-39) with an MTL whose diffuse map is a 1x1 PNG: texSample then returns the same texel
    whatever the uv, in the twin (texture per instance, render.h:243-245) and in the RTU shader (texture per material, closest.cpp:72-77);
  * twin scene: reference raycast Scene/BVH/TLAS (rcref_scene_create), instance i reflective as asked;
  * RTU scene: reference raytracing Scene + 4-wide quantised BVH (vxref_scene_create); the only edit is DATA: blas_node_t::reflectivity
    (offset 152) of each instance record is set to the same value the twin's instance has (the builder writes 0 there).
The rays are the RTU kernel's own camera rays (orc_generate_ray, pinned by tests/golden/camera_rays.npz); their radiance is what the
reference's Trace returns for them at max_depth 1..4 (rcref_radiance).  The fixture holds data only: the RTU-format buffers, the frame
size, the light, and per depth the f32 colours and RGB8 pixels."""
import os
import struct
import zlib

import numpy as np

from . import pyoracle as po

ASSETS = "/root/reference/tests/regression/raytracing/assets/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TMP = "/tmp/vx_golden_mirror"
LIGHT12 = np.array([120.0, 420.0, -160.0, 1.0, 1.0, 1.0, 0.35, 0.35, 0.35, 0.4, 0.35, 0.25], np.float32)   # light_pos, light_color, ambient, background


def png_1x1(path, rgb):
    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"\x00" + bytes(rgb)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def placed(name, tag, scale, shift, rgb):
    """The asset `name` with its vertices scaled by `scale` and moved by `shift`, an MTL with a 1x1 diffuse map of colour rgb."""
    os.makedirs(TMP, exist_ok=True)
    png = "c_%s.png" % tag
    png_1x1(os.path.join(TMP, png), rgb)
    with open(os.path.join(TMP, "m_%s.mtl" % tag), "w") as f:
        f.write("newmtl mm_%s\nKd %.6f %.6f %.6f\nKa 0.1 0.1 0.1\nmap_Kd %s\n" % (tag, rgb[0] / 256.0, rgb[1] / 256.0, rgb[2] / 256.0, png))
    out = ["mtllib m_%s.mtl\nusemtl mm_%s\n" % (tag, tag)]
    for line in open(ASSETS + name):
        if line.startswith("v "):
            x, y, z = [float(t) for t in line.split()[1:4]]
            line = "v %.6f %.6f %.6f\n" % (x * scale + shift[0], y * scale + shift[1], z * scale + shift[2])
        out.append(line)
    p = os.path.join(TMP, "%s_%s" % (tag, name))
    with open(p, "w") as f:
        f.write("".join(out))
    return p, os.path.join(TMP, png)


def make(tag, parts, w, h, depths=(1, 2, 3, 4)):
    """parts: [(asset, scale, shift, rgb, reflectivity)]"""
    objs, pngs, refl = [], [], []
    for i, (name, scale, shift, rgb, r) in enumerate(parts):
        o, p = placed(name, "%s%d" % (tag, i), scale, shift, rgb)
        objs.append(o); pngs.append(p); refl.append(r)
    twin = po.RefRcScene(objs, pngs, refl, rotate=False)
    rtu = po.ref_scene(objs)
    # the two builders place several meshes the same way (arrangeMeshesAroundY in both scene.cpp's): same instance transforms
    tb = twin.buffers["blas"].view(np.float32).reshape(len(parts), 40)
    rb = rtu["blas"].view(np.float32).reshape(len(parts), 40).copy()
    # (twin record: transform, invTransform, bvh_offset @128; RTU record: bvh_offset, invTransform, transform)
    assert np.array_equal(tb[:, 0:16], rb[:, 17:33]) and np.array_equal(tb[:, 16:32], rb[:, 1:17]), "instance transforms differ between the two reference builders"
    for i, r in enumerate(refl):          # blas_node_t::reflectivity @152 (both formats)
        rb[i, 38] = np.float32(r)
    rtu["blas"] = rb.view(np.uint8).reshape(-1).copy()
    assert np.array_equal(tb[:, 38], rb[:, 38])
    # (material_info_t::illum and ::reflectivity are left uninitialised by the reference's loader and read by nothing on this path --
    # the shader takes the reflectivity of the INSTANCE record: zeroed, so that the fixture is reproducible byte for byte)
    mat_dt = np.dtype([("f", "<f4", 16), ("tex_id", "<i4"), ("illum", "<i4"), ("tw", "<u4"), ("th", "<u4"), ("off", "<u8")])
    m = np.ascontiguousarray(rtu["mat"]).view(mat_dt).copy()
    m["illum"] = 0
    m["f"][:, 15] = 0.0
    rtu["mat"] = m.view(np.uint8).reshape(-1)
    rays = po.camera_rays(w, h)
    out = {k: rtu[k] for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    out.update(width=np.uint32(w), height=np.uint32(h), light12=LIGHT12, reflectivity=np.array(refl, np.float32), depths=np.array(depths, np.uint32))
    first = twin.trace(rays)
    out["hit_fraction"] = np.float32((first["dist"] < 1e29).mean())
    for d in depths:
        col, px = twin.radiance(rays, d, LIGHT12)
        out["colors_d%d" % d] = col.reshape(h, w, 3)
        out["rgb8_d%d" % d] = px.reshape(h, w)
    twin.close()
    np.savez_compressed(os.path.join(OUT, "mirror_%s.npz" % tag), **out)
    chg = [(int((out["rgb8_d%d" % a] != out["rgb8_d%d" % b]).sum())) for a, b in zip(depths, depths[1:])]
    print("mirror_%s: %dx%d, %.1f %% of the camera rays hit; pixels that change from one depth to the next: %s" % (tag, w, h, 100 * out["hit_fraction"], chg))


def main():
    # one reflective teapot (its spout, handle and lid mirror the body), seen from the RTU camera
    make("teapot", [("teapot.obj", 100.0, (130.0, 100.0, 0.0), (200, 100, 50), 0.5)], 120, 80)
    # the reference's own configuration (raycast/tracer.cpp:13,88-96): three copies of ONE model with reflectivity 0, 0.5, 0.3, placed by
    # arrangeMeshesAroundY in both builders.  (Copies of one model, not three models: render.h:72 indexes triIdx without the instance's
    # triangle offset, so in the reference's twin an instance of a second model is intersected with the first model's triangles.)
    make("trio", [("teapot.obj", 70.0, (200.0, 100.0, 0.0), (210, 210, 190), 0.0),
                  ("teapot.obj", 70.0, (200.0, 100.0, 0.0), (220, 40, 40), 0.5),
                  ("teapot.obj", 70.0, (200.0, 100.0, 0.0), (60, 120, 220), 0.3)], 128, 80)

if __name__ == "__main__":
    main()
