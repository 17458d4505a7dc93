#!/usr/bin/env python3
"""The 10 M-triangle hairball's primary rays (2,073,600 camera rays, lane utilisation 0.36 as whole tiles): the frame kernel (a tile of 64
pixels per wavefront) against the ray-buffer kernel (lanes refilled one by one) on the same rays."""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
rt = vrt.rtapi
sc = vrt.scene.procedural("hairball_fill", int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 250, 7)
ds = vrt.tracer.DeviceScene(sc, "cuda:0")
W, H = 1920, 1080
p = rt.default_shade_params()
p.light_pos[:] = (0.0, 400.0, 0.0)
px = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
ms_frame = timed(lambda: rt.render(ds.accel, W, H, 0, H, p, px.data_ptr(), 0, None, None, None, s))
rays = torch.zeros((W * H, 6), dtype=torch.float32, device="cuda:0")
L = rt._lib()
L.vxrt_camera_rays.restype = C.c_int
L.vxrt_camera_rays.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
assert L.vxrt_camera_rays(W, H, 0, H, rays.data_ptr(), s) == 0
hits = torch.zeros((W * H, 6), dtype=torch.int32, device="cuda:0")
ms_gen = timed(lambda: L.vxrt_camera_rays(W, H, 0, H, rays.data_ptr(), s))
ms_trace = timed(lambda: rt.trace(ds.accel, rays.data_ptr(), W * H, hits.data_ptr(), rt.MODE_CLOSEST, None, s))
print(json.dumps({"tris": sc.n_tris, "frame_kernel_primary_only_ms (traversal + shading)": round(ms_frame, 4), "camera_rays_ms": round(ms_gen, 4), "ray_buffer_kernel_ms": round(ms_trace, 4)}))
