#!/usr/bin/env python3
"""Tree quality anchored to the REFERENCE's builder (tests/regression/raytracing/bvh.cpp:30-264, through oracle/_ref -- checker code,
used here only to produce the tree that is measured): for the same triangles, the reference-built BVH4, the package's CPU SAH tree
(csrc/scene_builder.cpp) and, with a GPU, the tree of vxrt_bvh_build (csrc/bvh_builder.hip: Morton order, PLOC clustering, SAH-optimal 4-wide collapse) --
  * node / triangle fetches per ray and algorithmic bytes per ray (52 B per node, 36 B per triangle: SURVEY s8d) of the canonical
    traversal (oracle restatement, CPU) on a sample of the frame's camera rays,
  * with a GPU: Grays/s of the frame (primary + shadow, serial frames) on each tree through the same HIP kernels.
usage: python tests/tree_quality.py [--gpu] [--levels 6 8] [--fixtures teapot torus]"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (this harness lives under tests/: it runs the checker -- oracle/ -- which tools/ may not)
sys.path.insert(0, ROOT)
vrt = importlib.import_module("vortex-raytracing_amd")
from oracle import pyoracle as po   # noqa: E402


def write_obj(tri, path):
    v = tri.reshape(-1, 3)
    with open(path, "w") as f:
        f.write("\n".join("v %.9g %.9g %.9g" % (a, b, c) for a, b, c in v.tolist()))
        f.write("\n")
        n = len(tri)
        idx = np.arange(1, 3 * n + 1).reshape(n, 3)
        f.write("\n".join("f %d %d %d" % (a, b, c) for a, b, c in idx.tolist()))
        f.write("\n")


def fetches(scene, rays):
    _, st = po.trace_canonical(scene, rays)
    n = len(rays)
    nodes, tris = st["node_reads"] / n, st["tri_reads"] / n
    return {"node_fetches_per_ray": round(nodes, 3), "tri_fetches_per_ray": round(tris, 3), "bytes_per_ray": round(52 * nodes + 36 * tris, 1)}


def gpu_rate(scene_or_ds, w, h, light, frames=40):
    import torch
    ds = scene_or_ds if hasattr(scene_or_ds, "accel") else vrt.tracer.DeviceScene(scene_or_ds, "cuda:0")
    p = vrt.rtapi.default_shade_params()
    p.light_pos[:] = light
    px = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    vrt.rtapi.render(ds.accel, w, h, 0, h, p, px.data_ptr(), 1, None, None, cnt.data_ptr(), s)
    torch.cuda.synchronize()
    rays = int(cnt.item())
    for _ in range(10):
        vrt.rtapi.render(ds.accel, w, h, 0, h, p, px.data_ptr(), 1, None, None, None, s)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(frames):
        vrt.rtapi.render(ds.accel, w, h, 0, h, p, px.data_ptr(), 1, None, None, None, s)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / frames * 1e3
    st = vrt.rtapi.render_stats(ds.accel, w, h, 0, h, p, px.data_ptr(), 1, s)
    out = {"mrays_s_serial": round(rays / ms / 1e3, 1), "ms_per_frame": round(ms, 4), "frame_node_fetches_per_ray": round(st["node_fetches"] / st["rays"], 3),
           "frame_tri_fetches_per_ray": round(st["tri_fetches"] / st["rays"], 3), "pixels_crc": int(px.to(torch.int64).sum().item()),
           "accel_info_levels_shallow_ident_ldexp": [vrt.rtapi.accel_info(ds.accel, k) for k in range(4)],
           "frame_stats": {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in st.items() if isinstance(v, (int, float, np.integer, np.floating))}}
    ds.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--levels", type=int, nargs="*", default=[6, 8])
    ap.add_argument("--fixtures", nargs="*", default=["teapot", "torus"])
    ap.add_argument("--leaf-max", type=int, nargs="*", default=[0], help="vxrt_bvh_build's largest leaf (0 = its default); several = one GPU tree each")
    a = ap.parse_args()
    if not po.have_ref():
        raise SystemExit("oracle/_ref/libvxref.so (the reference's builder) is not built")
    for name in a.fixtures:      # the committed fixtures hold the reference-built buffers of the reference's own assets
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
        tri = g["tri"].view(np.float32).reshape(-1, 9)
        ours = vrt.scene.from_triangles([tri])
        row = {"scene": name, "tris": len(tri), "rays": "the fixture's %d rays" % len(g["rays"]),
               "reference_tree": fetches(g, g["rays"]), "cpu_sah_tree": fetches(ours, g["rays"])}
        if a.gpu:
            ds = vrt.tracer.DeviceScene.build_on_gpu(tri, None, None, None, "cuda:0")
            row["gpu_tree"] = fetches(ds.to_host(), g["rays"])
            ds.close()
        print(json.dumps(row), flush=True)
    LIGHT = (300.0, 480.0, 60.0)
    for lv in a.levels:
        sc = vrt.scene.procedural("atrium", lv, 0, 3)
        tri = sc["tri"].view(np.float32).reshape(-1, 9)
        d = tempfile.mkdtemp()
        path = os.path.join(d, "atrium.obj")
        write_obj(tri, path)
        t0 = time.time()
        ref = po.ref_scene([path])        # tinyobj + the reference's BVH builder and quantiser, host-compiled
        ref_s = time.time() - t0
        os.remove(path)
        w, h = 1920, 1080
        rays = po.camera_rays(w, h, 0, h)[:: 257 if lv >= 8 else 97]
        row = {"scene": "atrium level %d" % lv, "tris": len(tri), "rays": "%d camera rays spread over the 1920x1080 frame" % len(rays),
               "reference_builder_s": round(ref_s, 2), "reference_nodes": int(ref["bvh"].size // 52), "cpu_sah_nodes": sc.n_bvh_nodes,
               "reference_tree": fetches(ref, rays), "cpu_sah_tree": fetches(sc, rays)}
        a_, _ = po.trace_canonical(ref, rays)
        b_, _ = po.trace_canonical(sc, rays)
        row["same_distances"] = float((a_["dist"] == b_["dist"]).mean())
        if a.gpu:
            # the shading buffers of the reference-built scene: one untextured material (an OBJ without an MTL has none: give it the package's)
            refd = dict(ref)
            if refd["mat"].size == 0:
                refd["mat"], refd["tex"] = sc["mat"], sc["tex"]
                te = refd["triEx"].view(np.uint32).reshape(-1, 16).copy()
                te[:, 15] = 0
                refd["triEx"] = te.view(np.uint8).reshape(-1)
            row["reference_tree"].update(gpu_rate(refd, w, h, LIGHT))
            row["cpu_sah_tree"].update(gpu_rate(sc, w, h, LIGHT))
            ex = sc["triEx"].reshape(-1, 64)
            for lm in a.leaf_max:
                key = "gpu_tree" if lm == 0 else "gpu_tree_leaf_max_%d" % lm
                dsg = vrt.tracer.DeviceScene.build_on_gpu(tri, ex, sc["mat"], sc["tex"], "cuda:0", leaf_max=lm)
                row[key] = fetches(dsg.to_host(), rays)
                row[key].update({"nodes": int(dsg.bvh_info.n_nodes), "depth": int(dsg.bvh_info.max_depth)})
                row[key].update(gpu_rate(dsg, w, h, LIGHT))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
