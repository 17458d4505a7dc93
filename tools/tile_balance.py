"""Diagnostic: per-tile (per-wavefront) durations of one frame of the headline workload."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
W, H = 1920, 1080
sc = vrt.scene.procedural("atrium", 8, 0, 3)
ds = vrt.tracer.DeviceScene(sc, "cuda:0")
p = vrt.rtapi.default_shade_params(); p.light_pos[:] = (300.0, 480.0, 60.0)
px = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
for shadow in (0, 1):
    for _ in range(2):
        c = vrt.rtapi.render_stats(ds.accel, W, H, 0, H, p, px.data_ptr(), shadow, None, tile_clock=True)
    t = c["tile_clock"].astype(np.float64) / 100.0   # us
    dur = t[:, 1] - t[:, 0]
    t0 = t[:, 0].min(); span = t[:, 1].max() - t0
    print("shadow", shadow, "tiles", len(dur), "span_us %.1f" % span, "dur mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
    # concurrency over time
    edges = np.linspace(0, span, 25)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = (a + b) / 2 + t0
        live = int(((t[:, 0] <= mid) & (t[:, 1] > mid)).sum())
        print("   t=%6.0f us live waves %5d" % (mid - t0, live))
    # which rows are slow
    ty = (np.arange(len(dur)) // (W // 8))
    rows = [dur[ty == r].mean() for r in range(H // 8)]
    print("   mean tile dur by tile-row (every 9th):", " ".join("%.0f" % rows[r] for r in range(0, H // 8, 9)))
    order_start = np.argsort(t[:, 0])
    print("   start order vs tile id corr:", np.corrcoef(order_start, np.arange(len(dur)))[0, 1])
