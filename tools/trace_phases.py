#!/usr/bin/env python3
"""Where do the masked lanes of the ray-buffer kernel come from?  The counting build's per-wavefront log (vxrt_debug_trace_wave_log) of one
launch over the random rays of SURVEY s8d: loop iterations, runs of the node body / the leaf body and the lanes active in each.
usage: tools/trace_phases.py [n=4194304]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
dev = "cuda:0"
sc = vrt.scene.procedural("atrium", 8, 0, 3)
ds = vrt.tracer.DeviceScene(sc, dev)
g = torch.Generator(device=dev).manual_seed(12345)
lo, hi = torch.tensor(sc.bounds[:3], device=dev), torch.tensor(sc.bounds[3:], device=dev)
o = lo + (hi - lo) * torch.rand((n, 3), generator=g, device=dev)
d = torch.randn((n, 3), generator=g, device=dev)
rays = torch.cat([o, d / d.norm(dim=1, keepdim=True)], 1).contiguous()
hits = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
L = vrt.rtapi._lib()
L.vxrt_debug_trace_wave_log.restype = C.c_int
L.vxrt_debug_trace_wave_log.argtypes = [C.c_void_p, C.c_void_p]
log = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
assert L.vxrt_debug_trace_wave_log(ds.accel, log.data_ptr()) == 0
st = vrt.rtapi.trace_stats(ds.accel, rays.data_ptr(), n, hits.data_ptr(), vrt.rtapi.MODE_CLOSEST, None, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert L.vxrt_debug_trace_wave_log(ds.accel, None) == 0
raw = log.cpu().numpy(); raw[:, 9] &= (1 << 56) - 1
lg = raw[raw[:, 1] > 0].astype(np.float64)
it, nx, nl, lx, ll = (lg[:, k].sum() for k in range(3, 8))
tn, tl, tt, tf, tfin = (lg[:, k].sum() for k in (10, 11, 12, 13, 14))
print("%d rays, %d wavefronts; per ray: %.1f node fetches, %.1f triangle tests" % (n, len(lg), st["node_fetches"] / n, st["tri_fetches"] / n))
print("loop iterations per wavefront %.0f; node body: runs in %.3f of the iterations with %.1f of 64 lanes; leaf body: runs in %.3f with %.1f lanes" % (it / len(lg), nx / it, nl / max(nx, 1), lx / it, ll / max(lx, 1)))
print("lane steps per ray: node %.2f, leaf %.2f; wave-level body runs per 64 rays: node %.2f, leaf %.2f" % (nl / n, ll / n, nx * 64 / n, lx * 64 / n))
print("shader clocks of the wavefronts' lifetime: node body %.3f, instance + leaf part %.3f, fetch section %.3f, finish section %.3f, rest %.3f" %
      (tn / tt, tl / tt, tf / tt, tfin / tt, 1 - (tn + tl + tf + tfin) / tt))
print("clocks per node-body run %.0f, per leaf-part run %.0f, per iteration %.0f" % (tn / max(nx, 1), tl / max(lx, 1), tt / max(it, 1)))
