#!/usr/bin/env python3
"""The other BASELINE.json configurations, measured once each on one MI355X (they are parity-test cases, not the
bench line): prints one JSON object per configuration.  usage: python tools/config_bench.py [2] [3] [4] [5] [6] [--hair-strands N]   (6 = the raycast software twin, 7 = the vx_* call sequence, 8 = the GPU-built BLAS against the SAH tree)"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

vrt = importlib.import_module("vortex-raytracing_amd")
rtapi = vrt.rtapi
dev = "cuda:0"


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def render_cfg(tag, scene, W, H, light, steps):
    ds = vrt.tracer.DeviceScene(scene, dev)
    p = rtapi.default_shade_params()
    p.light_pos[:] = light
    px = torch.zeros((H, W), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rtapi.render(ds.accel, W, H, 0, H, p, px.data_ptr(), 1, None, None, cnt.data_ptr(), s)
    torch.cuda.synchronize()
    rays = int(cnt.item())
    ms = timed(lambda: rtapi.render(ds.accel, W, H, 0, H, p, px.data_ptr(), 1, None, None, None, s), steps)
    hit = float((px != px[0, 0]).float().mean().item())
    return {"config": tag, "tris": scene.n_tris, "resolution": [W, H], "rays_per_frame": rays, "ms_per_frame_serial": round(ms, 4),
            "mrays_s": round(rays / ms / 1e3, 1), "non_background_fraction": round(hit, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", type=int, default=[2, 3, 4, 5, 6, 7])
    ap.add_argument("--hair-strands", type=int, default=20000)   # x 250 segments x 2 = 10 M triangles
    a = ap.parse_args()
    out = []
    if 2 in a.configs:
        sc = vrt.scene.procedural("bunny", 6, 0, 1)    # 81,920-triangle "bunny-class" blob framed to fill the view
        out.append(render_cfg("configs[1]: bunny-class (framed), 1024x1024, primary + 1 shadow ray", sc, 1024, 1024, (20.0, 260.0, -150.0), 50))
    if 4 in a.configs:
        sc = vrt.scene.procedural("atrium", 8, 0, 3)
        out.append(render_cfg("configs[3] on one GPU: Sponza-class, 3840x2160, primary + 1 shadow ray (the 8-GPU split is bench.py --shard rows)", sc, 3840, 2160, (300.0, 480.0, 60.0), 20))
    if 5 in a.configs:
        cache = os.environ.pop("VXRT_SCENE_CACHE", None)     # (host_build_s is the build, not a load from the measurement scripts' cache)
        t0 = time.time()
        sc = vrt.scene.procedural("hairball_fill", a.hair_strands, 250, 7)    # framed to fill the 16:9 view
        build_s = time.time() - t0
        if cache:
            os.environ["VXRT_SCENE_CACHE"] = cache
        ds = vrt.tracer.DeviceScene(sc, dev)
        W, H, spp = 1920, 1080, 16
        b = sc.bounds
        radius = 0.25 * 0.5 * float(np.linalg.norm(np.array(b[3:]) - np.array(b[:3])))
        p = rtapi.default_shade_params()
        p.light_pos[:] = (0.0, 400.0, 0.0)
        px = torch.zeros((H, W), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        rtapi.render_ao(ds.accel, W, H, 0, H, p, spp, radius, px.data_ptr(), seed=7, rays_ptr=cnt.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert rtapi.status(s) == 0
        rays = int(cnt.item())
        ms = timed(lambda: rtapi.render_ao(ds.accel, W, H, 0, H, p, spp, radius, px.data_ptr(), seed=7, stream=s), 5, 1)
        out.append({"config": "configs[4]: hairball (framed), 1920x1080, 16 spp AO (tmax = 0.25 scene radius)", "tris": sc.n_tris, "bvh_nodes": sc.n_bvh_nodes,
                    "bvh_depth": sc.info.get("max_depth"), "host_build_s": round(build_s, 1), "rays_per_frame": rays,
                    "ms_per_frame": round(ms, 3), "mrays_s": round(rays / ms / 1e3, 1)})
    if 3 in a.configs:
        # configs[2] as worded ("1 bounce diffuse"): primary + one cosine-weighted closest-hit bounce per hit
        sc = vrt.scene.procedural("atrium", 8, 0, 3)
        ds = vrt.tracer.DeviceScene(sc, dev)
        W, H = 1920, 1080
        p = rtapi.default_shade_params()
        p.light_pos[:] = (300.0, 480.0, 60.0)
        px = torch.zeros((H, W), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        rtapi.render_diffuse_bounce(ds.accel, W, H, 0, H, p, px.data_ptr(), seed=3, rays_ptr=cnt.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert rtapi.status(s) == 0
        rays = int(cnt.item())
        ms = timed(lambda: rtapi.render_diffuse_bounce(ds.accel, W, H, 0, H, p, px.data_ptr(), seed=3, stream=s), 20)
        # the same with two frames in flight (two streams, two framebuffers), as the headline bench runs its frames
        rtapi.accel_frames_in_flight(ds.accel, 2)
        st2 = [torch.cuda.current_stream(), torch.cuda.Stream(device=dev)]
        px2 = [px, torch.zeros_like(px)]
        k = [0]
        def frame2():
            i = k[0] % 2
            k[0] += 1
            rtapi.render_diffuse_bounce(ds.accel, W, H, 0, H, p, px2[i].data_ptr(), seed=3, stream=st2[i].cuda_stream)
        for _ in range(4):
            frame2()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(40):
            frame2()
        torch.cuda.synchronize()
        ms2 = (time.time() - t0) / 40 * 1e3
        assert rtapi.status(s) == 0 and torch.equal(px2[0], px2[1])
        rtapi.accel_frames_in_flight(ds.accel, 1)
        out.append({"config": "configs[2] as worded: Sponza-class, 1920x1080, primary + 1 diffuse bounce (incoherent closest-hit rays)", "tris": sc.n_tris,
                    "rays_per_frame": rays, "ms_per_frame_serial": round(ms, 4), "mrays_s": round(rays / ms / 1e3, 1),
                    "ms_per_frame_2_in_flight": round(ms2, 4), "mrays_s_2_in_flight": round(rays / ms2 / 1e3, 1)})
    if 6 in a.configs:
        # software twin (tests/regression/raycast) on the same geometry in its own formats: BVH2, per-instance texture
        t0 = time.time()
        sc = vrt.scene.rc_procedural("atrium", 8, 0, 3)
        build_s = time.time() - t0
        ds = vrt.tracer.RcDeviceScene(sc, dev)
        W, H = 1920, 1080
        prm = rtapi.rc_params(vrt.scene.rc_camera_like_rtu(W, H), (300.0, 480.0, 60.0, 1, 1, 1, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25), 1, 1)
        px = torch.zeros((H, W), dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(30):      # (clocks, and the tile order the context learns from its own frames)
            rtapi.rc_render_accel(ds.accel, W, H, 0, H, prm, px.data_ptr(), None, s)
        ms = timed(lambda: rtapi.rc_render_accel(ds.accel, W, H, 0, H, prm, px.data_ptr(), None, s), 50)
        assert rtapi.status(s) == 0
        # two frames in flight: alternate streams and framebuffers (the accel keeps two frame contexts)
        st2 = [torch.cuda.current_stream(), torch.cuda.Stream(device=dev)]
        px2 = [px, torch.zeros_like(px)]
        def frame2(i):
            rtapi.rc_render_accel(ds.accel, W, H, 0, H, prm, px2[i % 2].data_ptr(), None, st2[i % 2].cuda_stream)
        for i in range(20):
            frame2(i)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(100):
            frame2(i)
        torch.cuda.synchronize()
        ms2 = (time.time() - t0) / 100 * 1e3
        assert rtapi.status(s) == 0 and torch.equal(px2[0], px2[1])
        out.append({"config": "software twin (raycast): Sponza-class BVH2, 1920x1080, primary rays, 1 spp", "tris": sc["tri"].size // 36,
                    "bvh2_nodes": sc["bvh"].size // 32, "bvh2_depth": sc["max_depth"], "host_build_s": round(build_s, 1),
                    "ms_per_frame": round(ms, 3), "mrays_s": round(W * H / ms / 1e3, 1),
                    "ms_per_frame_2_in_flight": round(ms2, 3), "mrays_s_2_in_flight": round(W * H / ms2 / 1e3, 1)})
    if 7 in a.configs:
        # the drop-in call sequence itself: vx_upload_bytes(kernel_arg) + vx_start + vx_ready_wait (+ vx_copy_from_dev), serial frames
        sc = vrt.scene.procedural("atrium", 8, 0, 3)
        W, H = 1920, 1080
        tr = vrt.tracer.Tracer(W, H)
        tr.init(sc)
        t0 = time.time()
        tr.setup(light_pos=(300.0, 480.0, 60.0), shadow=True)
        upload_s = time.time() - t0
        d = tr.dev
        tr.run()
        rays = d.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0)
        def frame(copy):
            args = d.upload_bytes(tr.kernel_arg)
            d.start(tr.krnl, args)
            d.ready_wait(vrt.runtime.VX_MAX_TIMEOUT)
            if copy:
                tr.bufs["out"].read()
            args.free()
        res = {}
        for copy in (False, True):
            for _ in range(5):
                frame(copy)
            t0 = time.time()
            n = 100
            for _ in range(n):
                frame(copy)
            res[copy] = (time.time() - t0) / n * 1e3
        tr.close()
        # the same call sequence from the C++ host (csrc/rt_host.cpp -N: what a compiled host like the reference's pays per frame; the ctypes host
        # above allocates a fresh 8 MB bytes object per vx_copy_from_dev and pays its page faults)
        cxx = {}
        try:
            import re, subprocess
            lib = vrt.LIB_DIR
            r = subprocess.run([os.path.join(lib, "rt_host"), "-m", "proc:atrium:8", "-w", str(W), "-h", str(H), "-S", "-L", "300,480,60", "-N", "1500", "-q", "-o", "/tmp/vxrt_cfg7.ppm",
                                "-k", os.path.join(vrt.VXBIN_DIR, "kernel.vxbin")], env=dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""), VORTEX_DRIVER="hip"),
                               capture_output=True, text=True, timeout=600)
            ms = [float(x) for x in re.findall(r"frame loop \(1500 frames, [^)]*\): ([0-9.]+) ms per frame", r.stdout)]
            if len(ms) == 2:
                cxx = {"cxx_host_ms_per_frame_start_wait": ms[0], "cxx_host_ms_per_frame_with_copy_from_dev": ms[1],
                       "cxx_host_mrays_s_start_wait": round(rays / ms[0] / 1e3, 1), "cxx_host_mrays_s_with_copy": round(rays / ms[1] / 1e3, 1)}
            # kernel_arg_t::samples_per_pixel = 5 (the reference host's `-s 5`: the frame's rays five times in ONE vx_start, kernel.cpp:67-80)
            r5 = subprocess.run([os.path.join(lib, "rt_host"), "-m", "proc:atrium:8", "-w", str(W), "-h", str(H), "-S", "-L", "300,480,60", "-s", "5", "-N", "300", "-q", "-o", "/tmp/vxrt_cfg7.ppm",
                                 "-k", os.path.join(vrt.VXBIN_DIR, "kernel.vxbin")], env=dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""), VORTEX_DRIVER="hip"),
                                capture_output=True, text=True, timeout=600)
            ms5 = [float(x) for x in re.findall(r"frame loop \(300 frames, [^)]*\): ([0-9.]+) ms per frame", r5.stdout)]
            if len(ms5) == 2:
                cxx.update({"cxx_host_spp5_ms_per_start_wait": ms5[0], "cxx_host_spp5_ms_per_start_with_copy_from_dev": ms5[1],
                            "cxx_host_spp5_mrays_s_start_wait": round(5 * rays / ms5[0] / 1e3, 1), "cxx_host_spp5_mrays_s_with_copy": round(5 * rays / ms5[1] / 1e3, 1)})
        except Exception as e:
            cxx = {"cxx_host_error": repr(e)[:200]}
        out.append({**cxx, "config": "drop-in vx_* sequence (ctypes host): Sponza-class, 1920x1080, primary + 1 shadow ray, serial frames", "rays_per_frame": int(rays),
                    "scene_upload_s": round(upload_s, 3), "ms_per_frame_start_wait": round(res[False], 4), "ms_per_frame_with_copy_from_dev": round(res[True], 4),
                    "mrays_s_start_wait": round(rays / res[False] / 1e3, 1), "mrays_s_with_copy": round(rays / res[True] / 1e3, 1)})
    if 8 in a.configs:
        # the BLAS built on the GPU (vxrt_bvh_build) against the CPU SAH builder: build time and what the tree is worth in the headline frame
        t0 = time.time()
        sc = vrt.scene.procedural("atrium", 8, 0, 3)
        sah_build_s = time.time() - t0
        tri = sc["tri"].view(np.float32).reshape(-1, 9)
        ex = sc["triEx"].reshape(-1, 64)
        perm = np.random.default_rng(1).permutation(len(tri))
        tri, ex = tri[perm].copy(), ex[perm].copy()
        W, H = 1920, 1080
        light = (300.0, 480.0, 60.0)
        base = render_cfg("SAH tree (CPU builder)", sc, W, H, light, 50)
        n = len(tri)
        s = torch.cuda.current_stream().cuda_stream
        for leaf_max in (1, 2, 4, 8):
            t_tri0, t_ex0 = torch.from_numpy(tri).to(dev), torch.from_numpy(ex).to(dev)
            nodes = torch.zeros(2 * n * 52, dtype=torch.uint8, device=dev)
            ts = []
            for _ in range(6):
                t_tri, t_ex = t_tri0.clone(), t_ex0.clone()
                torch.cuda.synchronize()
                t0 = time.time()
                info = rtapi.bvh_build(t_tri.data_ptr(), t_ex.data_ptr(), n, nodes.data_ptr(), 2 * n, 0, leaf_max, s)
                ts.append(time.time() - t0)
            del t_tri0, t_ex0, t_tri, t_ex, nodes
            ds = vrt.tracer.DeviceScene.build_on_gpu(tri, ex, sc["mat"], sc["tex"], dev, leaf_max=leaf_max)
            p = rtapi.default_shade_params()
            p.light_pos[:] = light
            px = torch.zeros((H, W), dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            rtapi.render(ds.accel, W, H, 0, H, p, px.data_ptr(), 1, None, None, cnt.data_ptr(), s)
            torch.cuda.synchronize()
            rays = int(cnt.item())
            ms = timed(lambda: rtapi.render(ds.accel, W, H, 0, H, p, px.data_ptr(), 1, None, None, None, s), 50)
            st = rtapi.render_stats(ds.accel, W, H, 0, H, p, px.data_ptr(), 1, s)
            out.append({"config": "GPU-built BLAS (vxrt_bvh_build, leaf_max %d): Sponza-class, 1920x1080, primary + 1 shadow ray, serial frames" % leaf_max,
                        "tris": n, "nodes": info.n_nodes, "leaves": info.n_leaves, "depth": info.max_depth,
                        "build_ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 2), "build_ms_min": round(min(ts) * 1e3, 2), "cpu_sah_build_s": round(sah_build_s, 2),
                        "rays_per_frame": rays, "ms_per_frame_serial": round(ms, 4), "mrays_s": round(rays / ms / 1e3, 1),
                        "node_fetches_per_ray": round(st["node_fetches"] / st["rays"], 2), "tri_fetches_per_ray": round(st["tri_fetches"] / st["rays"], 2),
                        "sah_tree_mrays_s": base["mrays_s"], "sah_tree_ms": base["ms_per_frame_serial"]})
            ds.close()
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
