#!/usr/bin/env python3
"""Item 8 probe: what does ordering the random-ray buffer buy the ray-buffer kernel (vxrt_trace), and what does the ordering cost?
Keys: Morton code of the origin's cell (b bits per axis) with the direction's octant / a quantised direction below it.  The sort and the
gather are torch's (rocPRIM radix sort) -- a bound on what a sort inside the library would cost."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
rtapi = vrt.rtapi
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16777216
scene = vrt.scene.procedural("atrium", 8, 0, 3)
ds = vrt.tracer.DeviceScene(scene, dev)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(12345)
lo = torch.tensor(scene.bounds[:3], device=dev); hi = torch.tensor(scene.bounds[3:], device=dev)
o = lo + (hi - lo) * torch.rand((n, 3), generator=g, device=dev)
d = torch.randn((n, 3), generator=g, device=dev); d = d / d.norm(dim=1, keepdim=True)
rays = torch.cat([o, d], 1).contiguous()
hits = torch.zeros(n * 24, dtype=torch.uint8, device=dev)


def rate(r, reps=5):
    for _ in range(2):
        rtapi.trace(ds.accel, r.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        rtapi.trace(ds.accel, r.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    assert rtapi.status(s) == 0
    return ms


def spread(v, bits):      # interleave: bit i of v -> bit 3 i
    out = torch.zeros_like(v)
    for i in range(bits):
        out |= ((v >> i) & 1) << (3 * i)
    return out


def key(bits, dirmode):
    q = ((o - lo) / (hi - lo) * (1 << bits)).clamp(0, (1 << bits) - 1).to(torch.int64)
    m = spread(q[:, 0], bits) | (spread(q[:, 1], bits) << 1) | (spread(q[:, 2], bits) << 2)
    if dirmode == "octant":
        k = (d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2)
        return (m << 3) | k, 3 * bits + 3
    if dirmode == "octant_first":
        k = (d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2)
        return (k << (3 * bits)) | m, 3 * bits + 3
    if dirmode == "dir6":    # 4 x 4 x 4 cells of the direction cube
        qd = ((d + 1) * 2).clamp(0, 3).to(torch.int64)
        k = qd[:, 0] | (qd[:, 1] << 2) | (qd[:, 2] << 4)
        return (m << 6) | k, 3 * bits + 6
    return m, 3 * bits


base = rate(rays)
print("unsorted: %.3f ms  %.1f Mrays/s" % (base, n / base / 1e3), flush=True)
ref = hits.clone()
for bits, dm in ((3, "none"), (4, "octant"), (5, "octant"), (6, "octant"), (5, "octant_first"), (5, "dir6"), (7, "dir6")):
    k, kb = key(bits, dm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        _, perm = torch.sort(k.to(torch.int32) if kb < 31 else k)
    torch.cuda.synchronize(); sort_ms = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        sr = rays[perm]
    torch.cuda.synchronize(); gather_ms = (time.perf_counter() - t0) / 3 * 1e3
    ms = rate(sr)
    h = hits.view(n, 24)
    t0 = time.perf_counter()
    for _ in range(3):
        back = torch.empty_like(h); back[perm] = h
    torch.cuda.synchronize(); scatter_ms = (time.perf_counter() - t0) / 3 * 1e3
    same = torch.equal(back.view(-1), ref)
    tot = ms + sort_ms + gather_ms + scatter_ms
    print("cells 2^%d/axis + %-12s (%2d key bits): trace %.3f ms (%.1f Mrays/s) | sort %.3f gather %.3f scatter %.3f -> total %.3f ms = %.1f Mrays/s; records equal after un-permuting: %s"
          % (bits, dm, kb, ms, n / ms / 1e3, sort_ms, gather_ms, scatter_ms, tot, n / tot / 1e3, same), flush=True)
