#!/usr/bin/env python3
"""Where the GPU-built tree loses against the CPU builder's: the 1920x1080 atrium frame with primary rays only and with the shadow rays,
serial frames, both trees; and the counting build's fetches of the TIMED traversal (occlusion rays unordered)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
sc = vrt.scene.procedural("atrium", 8, 0, 3)
p = vrt.rtapi.default_shade_params()
p.light_pos[:] = tuple(float(v) for v in os.environ.get("LIGHT", "300,480,60").split(","))
px = torch.zeros((1080, 1920), dtype=torch.int32, device="cuda:0")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for which in ("cpu", "gpu"):
    if which == "gpu":
        ds = vrt.tracer.DeviceScene.build_on_gpu(sc["tri"].view(np.float32).reshape(-1, 9), sc["triEx"].reshape(-1, 64), sc["mat"], sc["tex"], "cuda:0", leaf_max=int(os.environ.get("LEAFMAX", "2")))
    else:
        ds = vrt.tracer.DeviceScene(sc, "cuda:0")
    row = {"tree": which}
    for shadow in (0, 1):
        cnt.zero_()
        vrt.rtapi.render(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), shadow, None, None, cnt.data_ptr(), s)
        torch.cuda.synchronize()
        rays = int(cnt.item())
        for _ in range(10):
            vrt.rtapi.render(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), shadow, None, None, None, s)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(60):
            vrt.rtapi.render(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), shadow, None, None, None, s)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 60 * 1e3
        st = vrt.rtapi.render_stats(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), shadow, s, timed=True)
        sr = vrt.rtapi.render_stats(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), shadow, s, timed=False)
        row["shadow %d" % shadow] = {"ref_order_node_fetches": sr["node_fetches"], "ref_order_tri_fetches": sr["tri_fetches"], "ms": round(ms, 4), "rays": rays, "mrays_s": round(rays / ms / 1e3, 1), "timed_node_fetches": st["node_fetches"], "timed_tri_fetches": st["tri_fetches"]}
    print(json.dumps(row), flush=True)
    ds.close()
