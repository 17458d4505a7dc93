"""Diagnostic: when do the wavefronts of the persistent traversal launch start and stop?"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
W, H = 1920, 1080
sc = vrt.scene.procedural("atrium", 8, 0, 3)
ds = vrt.tracer.DeviceScene(sc, "cuda:0")
p = vrt.rtapi.default_shade_params(); p.light_pos[:] = (300.0, 480.0, 60.0)
px = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
L = vrt.rtapi._lib()
L.vxrt_render_wave_log.restype = C.c_int
L.vxrt_render_wave_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vrt.rtapi.ShadeParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for shadow in (1,):
    for it in range(3):   # the third frame runs with the tile order learned from the second
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        log = torch.zeros((4 * 8 * 256, 16), dtype=torch.int64, device="cuda:0")
        assert L.vxrt_render_wave_log(ds.accel, W, H, 0, H, C.byref(p), shadow, px.data_ptr(), cnt.data_ptr(), log.data_ptr(), None) == 0
        torch.cuda.synchronize()
    raw = log.cpu().numpy(); raw[:, 9] &= (1 << 56) - 1      # ([63:56] of entry 9: physical XCD)
    lg = raw.astype(np.float64)
    lg = lg[lg[:, 1] > 0]
    t0 = lg[:, 0].min()
    start, end, rays = (lg[:, 0] - t0) / 100.0, (lg[:, 1] - t0) / 100.0, lg[:, 2]
    span = end.max()
    print("waves", len(lg), "span_us %.1f" % span, "start p50 %.1f p99 %.1f max %.1f" % (np.percentile(start, 50), np.percentile(start, 99), start.max()))
    print("end: p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(end, q) for q in (1, 10, 50, 90, 100)))
    print("rays/wave: min %d p50 %d max %d" % (rays.min(), np.percentile(rays, 50), rays.max()))
    print("mean alive fraction %.3f" % ((end - start).sum() / (len(lg) * span)))
    it, nx, nl, lx, ll = (lg[:, k].sum() for k in range(3, 8))
    print("iterations/wave %.0f; node body: run in %.3f of iterations, %.1f lanes of 64; leaf body: run in %.3f, %.1f lanes"
          % (it / len(lg), nx / it, nl / max(nx, 1), lx / it, ll / max(lx, 1)))
    print("rays %d -> node steps/ray %.2f, leaf visits/ray %.2f; wave-level node runs/ray %.3f leaf runs/ray %.3f"
          % (rays.sum(), nl / rays.sum(), ll / rays.sum(), nx * 64 / rays.sum(), lx * 64 / rays.sum()))
    tn, tl, tt = lg[:, 10].sum(), lg[:, 11].sum(), lg[:, 12].sum()
    print("shader clocks: node body %.3f, instance+leaf part %.3f, rest (fetch, finish, loop control) %.3f of the wave lifetime" % (tn / tt, tl / tt, 1 - (tn + tl) / tt))
    print("node steps served from the LDS top-of-tree image: %.3f of lane node steps; node-body runs with every node lane at the SAME node: %.3f"
          % (lg[:, 8].sum() / max(nl, 1), lg[:, 9].sum() / max(nx, 1)))
    print("shader clocks: fetch section %.3f, finish section %.3f of the wave lifetime" % (lg[:, 13].sum() / tt, lg[:, 14].sum() / tt))
    print("shader clocks per node-body run %.0f, per instance+leaf run %.0f, per iteration %.0f" % (tn / max(nx, 1), tl / max(lx, 1), tt / max(it, 1)))
