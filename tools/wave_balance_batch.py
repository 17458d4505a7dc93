"""Diagnostic: the traversal launch of one rank's SET of frames (vxrt_render_interleaved_batch) seen per wavefront -- when does the queue
run dry, when do the wavefronts end, how long is the tail?  usage: tools/wave_balance_batch.py [world=8] [frames=10] [rank=0]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 1920, 1080
sc = vrt.scene.procedural("atrium", 8, 0, 3)
ds = vrt.tracer.DeviceScene(sc, "cuda:0")
p = vrt.rtapi.default_shade_params(); p.light_pos[:] = (300.0, 480.0, 60.0)
ig = vrt.sharding.InterleavedGather(H, W, rank, world, "cuda:0", slots=1, collective=False, batch=nf)
buf = ig.new_frame_buffer("cuda:0")
L = vrt.rtapi._lib()
fn = L.vxrt_render_interleaved_batch_wave_log
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_uint32] * 5 + [C.POINTER(vrt.rtapi.ShadeParams), C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
arr = (vrt.rtapi.ShadeParams * nf)(*([p] * nf))
vrt.rtapi.accel_frames_in_flight(ds.accel, 2)       # (the bench's setting: batches then take the learned order)
for it in range(3):      # the third set runs with the tile order learned from the second
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    log = torch.zeros((4 * 8 * 256, 16), dtype=torch.int64, device="cuda:0")
    assert fn(ds.accel, W, H, rank, world, nf, arr, 1, buf.data_ptr(), ig.frame_stride, cnt.data_ptr(), log.data_ptr(), None) == 0
    torch.cuda.synchronize()
    raw = log.cpu().numpy()
    live = raw[:, 1] > 0
    xcd = ((raw[:, 9].astype(np.uint64) >> np.uint64(56)) & np.uint64(15)).astype(np.int64)[live]     # physical XCD of the wavefront (HW_REG_XCC_ID)
    grp = ((np.arange(raw.shape[0]) // 4) % 8)[live]                                                  # blockIdx % 8 of the wavefront's block
    raw[:, 9] &= (1 << 56) - 1
    lg = raw.astype(np.float64)[live]
    if it == 2:
        print("   blockIdx %% 8 -> physical XCD of its wavefronts:", {int(g): sorted(set(xcd[grp == g].tolist())) for g in range(8)})
    t0 = lg[:, 0].min()
    start, end, rays, tq = (lg[:, 0] - t0) / 100.0, (lg[:, 1] - t0) / 100.0, lg[:, 2], (lg[:, 15] - t0) / 100.0
    span = end.max()
    tiles = rays / 64.0
    print("set %d: waves %d span_us %.1f; start p50 %.1f p99 %.1f max %.1f" % (it, len(lg), span, np.percentile(start, 50), np.percentile(start, 99), start.max()))
    print("   queue found empty: first %.1f p50 %.1f last %.1f us" % (tq[tq > 0].min(), np.percentile(tq[tq > 0], 50), tq.max()))
    print("   end: p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, q) for q in (1, 10, 50, 90, 99, 100)))
    print("   last tile of a wavefront (end - queue empty): p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile((end - tq)[tq > 0], q) for q in (50, 90, 99, 100)))
    print("   rays/wave: min %d p50 %d max %d; mean alive fraction %.3f; total rays %d" % (rays.min(), np.percentile(rays, 50), rays.max(), (end - start).sum() / (len(lg) * span), rays.sum()))
    it_, nx, nl, lx, ll = (lg[:, k].sum() for k in range(3, 8))
    print("   node body: %.1f lanes of 64, leaf body %.1f lanes; iterations/wave %.0f" % (nl / max(nx, 1), ll / max(lx, 1), it_ / len(lg)))
    # shader clock each wavefront saw (s_memtime cycles of its lifetime / its lifetime on the constant 100 MHz clock), and where its time went, by end time
    ghz = lg[:, 12] / np.maximum(lg[:, 1] - lg[:, 0], 1.0) / 10.0
    early, late = end <= np.percentile(end, 10), end >= np.percentile(end, 90)
    for name, m_ in (("first 10 % to end", early), ("last 10 % to end", late)):
        print("   %-18s: shader clock %.3f GHz; shader clocks per iteration %.0f; fetch section %.3f, finish section %.3f of the lifetime; iterations %.0f, tiles %.1f" %
              (name, ghz[m_].mean(), lg[m_, 12].sum() / max(lg[m_, 3].sum(), 1), lg[m_, 13].sum() / lg[m_, 12].sum(), lg[m_, 14].sum() / lg[m_, 12].sum(), lg[m_, 3].mean(), tiles[m_].mean() / 2))
        print("   %-18s  clocks per node-body run %.0f, per leaf-body run %.0f; per wavefront: node+leaf bodies %.0f us, fetch %.0f us, finish %.0f us, rest %.0f us, dry polls %.0f us" %
              ("", lg[m_, 10].sum() / max(lg[m_, 4].sum(), 1), lg[m_, 11].sum() / max(lg[m_, 6].sum(), 1),
               ((lg[m_, 10] + lg[m_, 11]) / ghz[m_] / 1e3).mean(), (lg[m_, 13] / ghz[m_] / 1e3).mean(), (lg[m_, 14] / ghz[m_] / 1e3).mean(),
               ((lg[m_, 12] - lg[m_, 10] - lg[m_, 11] - lg[m_, 13] - lg[m_, 14]) / ghz[m_] / 1e3).mean(), (lg[m_, 8] / ghz[m_] / 1e3).mean()))
    print("   reservations that met a dry shard: %.1f us per wavefront on average, p90 %.1f, max %.1f (shader clocks at the wavefront's clock)" %
          ((lg[:, 8] / np.maximum(ghz, 0.1) / 1e3).mean(), np.percentile(lg[:, 8] / np.maximum(ghz, 0.1) / 1e3, 90), (lg[:, 8] / np.maximum(ghz, 0.1) / 1e3).max()))
    for x in range(8):
        m_ = xcd == x
        print("   physical XCD %d (home band %d): end p10 %.0f p50 %.0f p90 %.0f us; iterations %.0f; clocks per iteration %.0f; node-body run %.0f clocks; tiles %.1f" %
              (x, x, *np.percentile(end[m_], (10, 50, 90)), lg[m_, 3].mean(), lg[m_, 12].sum() / max(lg[m_, 3].sum(), 1), lg[m_, 10].sum() / max(lg[m_, 4].sum(), 1), tiles[m_].mean() / 2))
    # how the tail is populated: wavefronts still alive at fractions of the span
    for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
        print("   alive at %.2f of the span: %d wavefronts" % (f, int(((start <= f * span) & (end > f * span)).sum())), end=";")
    print()
