import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
rtapi = vrt.rtapi
for name, a, b, seed in (("hairball_fill", 2000, 250, 7), ("hairball_fill", 20000, 250, 7), ("bunny", 6, 0, 1)):
    t0 = time.time(); sc = vrt.scene.procedural(name, a, b, seed); cpu_s = time.time() - t0
    tri = sc["tri"].view(np.float32).reshape(-1, 9); ex = sc["triEx"].reshape(-1, 64)
    n = len(tri)
    perm = np.random.default_rng(1).permutation(n)
    tri, ex = tri[perm].copy(), ex[perm].copy()
    for leaf_max in (2, 4):
        try:
            t0 = time.time()
            ds = vrt.tracer.DeviceScene.build_on_gpu(tri, ex, sc["mat"], sc["tex"], "cuda:0", leaf_max=leaf_max)
            print(name, n, "leaf_max", leaf_max, "nodes", ds.bvh_info.n_nodes, "depth", ds.bvh_info.max_depth, "cpu depth", sc.info["max_depth"], "cpu build s %.2f" % cpu_s, flush=True)
        except Exception as e:
            print(name, n, "leaf_max", leaf_max, "FAILED", repr(e), flush=True); continue
        W, H = 1920, 1080
        p = rtapi.default_shade_params(); p.light_pos[:] = (0.0, 400.0, 0.0)
        px = torch.zeros((H, W), dtype=torch.int32, device="cuda:0"); px2 = torch.zeros_like(px)
        s = torch.cuda.current_stream().cuda_stream
        dr = vrt.tracer.DeviceScene(sc, "cuda:0")
        for d, o in ((ds, px), (dr, px2)):
            rtapi.render(d.accel, W, H, 0, H, p, o.data_ptr(), 1, None, None, None, s); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): rtapi.render(d.accel, W, H, 0, H, p, o.data_ptr(), 1, None, None, None, s)
            e1.record(); torch.cuda.synchronize()
            print("   ", "gpu tree" if d is ds else "sah tree", "ms/frame %.3f" % (e0.elapsed_time(e1) / 10), "status", rtapi.status(s), flush=True)
        print("    pixels equal fraction %.5f" % float((px == px2).float().mean()), flush=True)
        ds.close(); dr.close()
