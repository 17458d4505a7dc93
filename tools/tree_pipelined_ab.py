#!/usr/bin/env python3
"""The headline's pipelined schedule (sets of 5 frames, 2 sets in flight: what bench.py times) on the CPU builder's tree and on the tree
vxrt_bvh_build makes of the same triangles -- what switching the bench's scene to the GPU builder would be worth."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
rt = vrt.rtapi
sc = vrt.scene.procedural("atrium", 8, 0, 3)
W, H = 1920, 1080
p = rt.default_shade_params()
p.light_pos[:] = (300.0, 480.0, 60.0)
px = [torch.zeros((5, H, W), dtype=torch.int32, device="cuda:0") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
for rep in range(2):
    for which in ("cpu", "gpu"):
        if which == "gpu":
            ds = vrt.tracer.DeviceScene.build_on_gpu(sc["tri"].view(np.float32).reshape(-1, 9), sc["triEx"].reshape(-1, 64), sc["mat"], sc["tex"], "cuda:0", leaf_max=2)
        else:
            ds = vrt.tracer.DeviceScene(sc, "cuda:0")
        rt.accel_frames_in_flight(ds.accel, 2)
        cnt.zero_()
        rt.render(ds.accel, W, H, 0, H, p, px[0].data_ptr(), 1, None, None, cnt.data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize()
        rays = int(cnt.item())
        def run(sets):
            for i in range(sets):
                rt.render_batch(ds.accel, W, H, [p] * 5, px[i % 2].data_ptr(), W * H, 1, None, streams[i % 2].cuda_stream)
        run(40)
        torch.cuda.synchronize()
        t0 = time.time()
        run(40)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 200 * 1e3
        # random rays (origins in the scene's box, directions on the sphere), closest hit, 8 Mi
        g = torch.Generator(device="cuda:0"); g.manual_seed(12345)
        nrr = 8 << 20
        lo = torch.tensor(sc.bounds[:3], device="cuda:0"); hi = torch.tensor(sc.bounds[3:], device="cuda:0")
        o = lo + (hi - lo) * torch.rand((nrr, 3), device="cuda:0", generator=g)
        d = torch.randn((nrr, 3), device="cuda:0", generator=g); d = d / d.norm(dim=1, keepdim=True)
        rr = torch.cat([o, d], 1).contiguous()
        hits = torch.zeros((nrr, 6), dtype=torch.int32, device="cuda:0")
        s0 = streams[0].cuda_stream
        rt.trace(ds.accel, rr.data_ptr(), nrr, hits.data_ptr(), rt.MODE_CLOSEST, None, s0); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5): rt.trace(ds.accel, rr.data_ptr(), nrr, hits.data_ptr(), rt.MODE_CLOSEST, None, s0)
        torch.cuda.synchronize()
        ms_rr = (time.time() - t0) / 5 * 1e3
        print(json.dumps({"tree": which, "ms_per_frame_pipelined": round(ms, 4), "mrays_s": round(rays / ms / 1e3, 1), "random_rays_mrays_s": round(nrr / ms_rr / 1e3, 1)}), flush=True)
        del rr, hits, o, d
        ds.close()
