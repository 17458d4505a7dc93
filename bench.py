#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ray-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow) at 1920x1080 on the 1M-triangle "Sponza-class" BVH.  A step = one frame of
the RTU test on that scene: camera ray -> closest hit -> Lambert shade with one occlusion ray toward the light per hit -> RGB8,
through the C ABI (persistent traversal launch + EXACT launches + shading pass), scene already resident in HBM.

How the K timed steps are issued (all of it inside the timed region, every ray of every frame traced):
  * frames go out in equal groups of at most 5 per set of launches (vxrt_render_batch: each wavefront then works through five
    times as many tiles per launch, so the launch's ramp and tail weigh less: +8 %; --batch 1 = one frame per set of launches);
  * sets are issued round robin on --frames-in-flight streams (default 2; 2 > 4 > 3 measured), so the draining tail of one set's
    persistent launch overlaps the next set's; --frames-in-flight 1 gives strictly serial single frames;
  * before the W warmup steps, --settle-frames untimed frames of the same kind bring the GPU clocks to their sustained state (a
    20-step run would otherwise time the clock ramp: DESIGN.md s10).

Multi-GPU (one process per GPU, torch.distributed over RCCL): ONE frame is split by 8-row tile rows of the reference grid
(kernel.cpp:128-133) -- rank r renders the tile rows r, r + N, r + 2N, ... (interleaving balances the ranks by construction: the
cost of a tile varies 4x over the frame) with no data-path collective; frames are traced in sets of up to 16 per set of launches
(vxrt_render_interleaved_batch: a rank's share of one frame is too small to fill its GPU) and ONE gather over xGMI per set
assembles the images on rank 0 on its own stream (north_star: "RCCL gather only for final image assembly").  STRONG scaling:
the frame, and so the total work, is fixed as N grows; value = rays of the whole frame x K / max-over-ranks time.
`--shard bands` (built and measured in round 4, not the default): N contiguous bands cut on tile boundaries where the measured
COST is equal (planned from the ranks' own frame times during the untimed settle phase, sharding.rebalance_bands), rank 0
rendering its band in place in the final images and receiving every other band straight into its place -- one group of
ncclSend / ncclRecv per set, no extraction or interleaving copy -- with the sets of the timed steps tapering (20 -> 10, 5, 2, 2, 1).
It loses on both counts: a band of the expensive part of the frame is 11 tile rows high at N = 8, so the cut cannot
be placed finer than 9 % of a rank's work, and a set of one or two frames' shares is a single round of tiles that takes as long
as ten.  `--shard rows` = equal-height bands, one frame per set of launches.  `--rehearse-world N` does on ONE GPU what rank 0 of
N would do, without the collective (diagnostic).

Prints one JSON line (driver contract) with `roofline` and `cpu_baseline` objects.  The roofline is the VALU roof: this
traversal is cache-resident pointer chasing whose binding resource is vector-ALU issue, priced with the
per-opcode SIMD cycles measured by tools/calibrate_valu.py; the algorithmic-bytes figure of SURVEY s8d is reported against the
HBM and L2 peaks next to it, with the measured HBM traffic.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
L2_PEAK_GBS = 34500.0   # aggregate L2 (same guide): the level that serves this cache-resident scene
SIMDS = 1024            # 256 CUs x 4
CLOCK_GHZ = 2.4         # nominal shader clock; the calibration loops held 2.2-2.4 GHz under load


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--level", type=int, default=8, help="atrium tessellation level; 8 -> 1,048,576 triangles")
    ap.add_argument("--shard", choices=["tilerows", "bands", "rows"], default="tilerows",
                    help="how ONE frame is split over the GPUs: interleaved 8-row tile rows, one gather + one interleaving copy per set (default); contiguous bands cut at equal measured cost, received in place (measured slower: DESIGN.md s6); equal-height bands, one frame per launch")
    ap.add_argument("--taper", default="auto",
                    help="--shard bands: sizes of the sets the timed steps are issued in, e.g. 10,5,3,2 (sum = --steps); auto = sharding.taper(steps); none = equal sets as --batch decides")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per set of launches, and per collective with several GPUs (0 = automatic: up to 5 on one GPU, up to 16 with several, the timed steps split into equal groups; 1 = one frame per set of launches)")
    ap.add_argument("--sets", default="auto",
                    help="several GPUs, --shard tilerows: sizes of the sets of frames the timed steps are issued in, e.g. 8,8,4 (sum = --steps; each 1..32); "
                         "auto = what --batch decides (equal sets), tapered at the end when the run is long enough (see main)")
    ap.add_argument("--wire", choices=["auto", "rgb24", "rgba32"], default="auto",
                    help="several GPUs, --shard tilerows: a share's pixels on the link as 3 bytes (0x00RRGGBB without its zero byte; default where the width is a multiple of 4) or as they are")
    ap.add_argument("--no-leg-4k", dest="leg_4k", action="store_false",
                    help="several GPUs: skip the second leg (BASELINE configs[3]: the same scene at 3840x2160 split over the ranks, reported under extras.leg_3840x2160)")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="diagnostic, one GPU: do per frame what rank 0 of an N-GPU run does (its share of the frame, extraction, assembly) without the collective")
    ap.add_argument("--no-shadow", action="store_true")
    ap.add_argument("--settle-frames", type=int, default=160,
                    help="untimed frames before the warmup steps that bring the GPU clocks to their sustained state (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU-seconds of host work for the main cpu_baseline leg")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="frames kept in flight on as many HIP streams (vxrt_accel_frames_in_flight); 1 = strictly serial frames")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal of the N>1 code path where ranks share one GPU (shares gathered through host memory)")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="diagnostic, one GPU: run the N > 1 code path (process group, interleaved batches, dist.gather, barriers, all_reduce) with a group of ONE rank -- the RCCL calls of the multi-GPU run on a 1-GPU box")
    ap.add_argument("--other-configs", choices=["none", "short", "full"], default="full",
                    help="1 GPU only: short host-timed legs of BASELINE.json's other configurations under extras.other_configs (short: bunny-class 1024^2, "
                         "the diffuse bounce, 3840x2160; full, the default: + configs[4], the 10M-triangle hairball with 16 spp AO -- 3 timed frames; its tree "
                         "takes ~6 s of host time to build)")
    ap.add_argument("--other-configs-child", choices=["short", "full"], default=None, help=argparse.SUPPRESS)
    ap.add_argument("--random-rays", type=int, default=16777216,
                    help="second leg (SURVEY s8d 'random rays vs fixed BVH'): N incoherent rays per GPU through vxrt_trace, reported under extras; 0 = skip")
    return ap.parse_args()


def cpu_baseline(scene, vrt, w, h, light, budget_cpu_s, random_sample=None):
    """BASELINE.md s2, timed on this box's host cores, on BOUNDED samples of the same workload.  Checker code (oracle/_ref, or
    the C restatement if that library is absent) is the thing timed here for the reported baseline only, never for `value`.
      B2-N  (value)  the reference's own BVHTraverser (sim/simx/rt_traversal.cpp via oracle/_ref) on the frame's camera rays,
                     closest hit, one traverser per thread over disjoint ray ranges, all host cores
      B2-1           the same, one thread
      B1             the reference's software ray caster (raycast render.h GenerateRay + Trace via oracle/_ref), one thread as
                     written, render loop only, on the same geometry as BVH2 (primary rays + Lambert shade)"""
    import concurrent.futures as cf
    import numpy as np
    from oracle import pyoracle as po
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))   # the box's CPU share for one GPU is 16 cores
    rays = po.camera_rays(w, h)
    img = po.Image(scene)

    def run(fn, rays, repeats, threads):
        chunks = np.array_split(np.arange(len(rays)), threads * 8)
        t0 = time.perf_counter()
        done = 0
        with cf.ThreadPoolExecutor(threads) as ex:   # ctypes releases the GIL during the foreign call
            futs = [ex.submit(fn, img, rays[c]) for _ in range(repeats) for c in chunks]
            for f in futs:
                try:
                    out, _ = f.result(timeout=max(5.0, 8 * budget_cpu_s / threads - (time.perf_counter() - t0)))
                    done += len(out)
                except cf.TimeoutError:
                    return None
        return done, time.perf_counter() - t0

    def sized(fn, budget):
        t0 = time.perf_counter()
        fn(img, rays[:: max(1, len(rays) // 4000)])          # probe spread over the frame
        per_ray = (time.perf_counter() - t0) / len(rays[:: max(1, len(rays) // 4000)])
        want = budget / per_ray                              # rays for the CPU-second budget
        if want >= len(rays):
            return rays, max(1, int(round(want / len(rays))))
        return rays[:: max(1, int(len(rays) / want))], 1

    kind, res, fn = "port", None, po.trace_faithful
    if po.have_ref():
        r, rep = sized(po.trace_ref, budget_cpu_s)
        res = run(po.trace_ref, r, rep, cores)
        if res:
            kind, fn = "reference", po.trace_ref
    if res is None:
        r, rep = sized(po.trace_faithful, budget_cpu_s)
        res = run(po.trace_faithful, r, rep, cores)
    done, dt = res
    what = "reference sim/simx/rt_traversal.cpp via oracle/_ref" if kind == "reference" else "oracle/rt_oracle.c restatement"
    out = {"value": round(done / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": kind,
           "sample": "B2-N: %d primary camera rays of the %dx%d frame (closest hit, no shadow rays; %d x %d rays), %s, %d host threads, %.1f s wall"
                     % (done, w, h, rep, len(r), what, cores, dt)}
    # B2-1: one thread, 3 CPU-seconds
    r1, rep1 = sized(fn, 3.0)
    res1 = run(fn, r1, rep1, 1)
    if res1:
        out["b2_single_thread"] = {"value": round(res1[0] / res1[1] / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                                   "sample": "%d rays (every %d-th camera ray of the frame), %.1f s wall" % (res1[0], max(1, len(rays) // max(1, len(r1))), res1[1])}
    # B1: the reference's raycast render loop on the same geometry as BVH2, rows spread over the frame, one thread
    if po.have_ref_rc():
        try:
            rc = vrt.scene.rc_procedural("atrium", int(round(np.log2(scene.n_tris) / 2 - 2)), 0, 3)
            cam = vrt.scene.rc_camera_like_rtu(w, h)
            light12 = tuple(light) + (1.0, 1.0, 1.0, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25)
            t0 = time.perf_counter()
            n = 0
            stride, off = max(1, h // 64), 0
            while time.perf_counter() - t0 < 3.0 and off < stride:   # one full row at a time, each pass spread over the whole frame
                for y in range(off, h, stride):
                    _, k = po.ref_rc_render_buffers(rc, w, h, y, y + 1, cam, light12)
                    n += k
                    if time.perf_counter() - t0 >= 3.0:
                        break
                off += 1
            dt1 = time.perf_counter() - t0
            out["b1_raycast_render_loop"] = {"value": round(n / dt1 / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                                             "sample": "%d primary rays (+ shade) of %d rows spread over the %dx%d frame, reference raycast render.h via oracle/_ref on the BVH2 of the same %d triangles, %.1f s wall"
                                                       % (n, n // w, w, h, rc["tri"].size // 36, dt1)}
        except Exception as e:   # the baseline leg must not take the bench line down
            out["b1_raycast_render_loop"] = {"error": repr(e)[:200]}
    # like for like with the GPU run (BASELINE.md s2 "the same synthetic rays as the GPU run"), reference object code, all cores:
    #   * the frame's own mix -- primary rays of rows spread over the frame AND the occlusion ray of every hit toward the light
    #     (first accepted candidate = the any-hit query the GPU's shadow phase answers), constructed as shadow_ray does
    #   * a sample of the second leg's random rays (the device-generated buffer itself, copied back)
    if kind == "reference":
        try:
            f = np.float32
            rows = np.arange(0, h, 4)
            pr = np.ascontiguousarray(rays.reshape(h, w, 6)[rows].reshape(-1, 6))
            t0 = time.perf_counter()
            chunks = np.array_split(np.arange(len(pr)), cores * 4)
            with cf.ThreadPoolExecutor(cores) as ex:
                hits = np.concatenate([x[0] for x in ex.map(lambda c: po.trace_ref(img, pr[c]), chunks)])
            hit = hits["dist"] < 1e29
            ph, dh = pr[hit], hits["dist"][hit].reshape(-1, 1).astype(f)      # (only the rays that hit: a miss's 1e30 would overflow the squares below)
            I = (ph[:, :3] + ph[:, 3:] * dh).astype(f)
            L = (np.array(light, f)[None] - I).astype(f)
            dist = np.sqrt((L[:, 0] * L[:, 0] + L[:, 1] * L[:, 1]).astype(f) + (L[:, 2] * L[:, 2]).astype(f)).astype(f)
            Ln = (L * (f(1.0) / dist)[:, None]).astype(f)
            sr = np.concatenate([(I + (Ln * f(0.001)).astype(f)).astype(f), Ln], 1).astype(f)
            schunks = np.array_split(np.arange(len(sr)), cores * 4)
            with cf.ThreadPoolExecutor(cores) as ex:     # (the reference traverser has no tmax: an any-hit query to infinity, an upper bound of the GPU's bounded one)
                list(ex.map(lambda c: po.trace_ref(img, sr[c], any_hit=True), schunks))
            dt2 = time.perf_counter() - t0
            out["same_rays_primary_plus_shadow"] = {"value": round((len(pr) + len(sr)) / dt2 / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "reference",
                                                    "sample": "%d primary + %d occlusion rays of %d rows spread over the %dx%d frame, %.1f s wall (ray construction included)" % (len(pr), len(sr), len(rows), w, h, dt2)}
        except Exception as e:
            out["same_rays_primary_plus_shadow"] = {"error": repr(e)[:200]}
        if random_sample is not None and len(random_sample):
            try:
                rr = np.ascontiguousarray(random_sample, np.float32)
                t0 = time.perf_counter()
                chunks = np.array_split(np.arange(len(rr)), cores * 4)
                with cf.ThreadPoolExecutor(cores) as ex:
                    list(ex.map(lambda c: po.trace_ref(img, rr[c]), chunks))
                dt3 = time.perf_counter() - t0
                out["same_rays_random"] = {"value": round(len(rr) / dt3 / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "reference",
                                           "sample": "the first %d of the GPU leg's random rays (seed 12345, origins in the scene box, directions on the sphere), closest hit, %.1f s wall" % (len(rr), dt3)}
            except Exception as e:
                out["same_rays_random"] = {"error": repr(e)[:200]}
    out["note"] = "no POCL path exists to time: the reference has no OpenCL ray tracer and POCL is not installed (SURVEY s0.5)"
    return out


class ClockProbe:
    """Shader clock the chip holds around the timed region: a one-wavefront kernel of the measurement library (csrc/calib_kernels.hip,
    lib/libvxrt_calib.so -- not the product library) reads s_memtime and the constant 100 MHz s_memrealtime around a ~30 us spin.
    Launched on the timed region's first stream right before its first step and right after its last one: the power management
    moves the clock over milliseconds, the probe sits within microseconds of the load."""

    def __init__(self, vrt, torch, dev):
        import ctypes as C
        self.ok = False
        try:
            self.L = C.CDLL(vrt.lib_path("libvxrt_calib.so"))
            self.L.vxcal_clock_probe.restype = C.c_int
            self.L.vxcal_clock_probe.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
            self.buf = torch.zeros((4, 4), dtype=torch.int64, device=dev)
            self.ok = True
        except Exception as e:     # the probe is optional: without it the nominal clock is used, and the line says so
            self.err = repr(e)[:120]

    def launch(self, slot, stream_ptr):
        if self.ok and self.L.vxcal_clock_probe(3000, self.buf[slot].data_ptr(), stream_ptr) != 0:
            self.ok = False

    def ghz(self, slot):
        if not self.ok:
            return None
        c, t = int(self.buf[slot, 0].item()), int(self.buf[slot, 1].item())
        return (c / t / 10.0) if t > 0 else None


def load_profile_constants():
    """Per-frame counters of the timed kernel that a bench run cannot read itself (rocprofv3 --pmc needs its own passes):
    profiles/valu_profile.json, written by tools/roofline_from_pmc.py from the round's PMC summary.  Labelled as profile
    constants wherever they are used."""
    p = os.path.join(ROOT, "profiles", "valu_profile.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def kernel_source_id():
    """16 hex digits that change when the code of the timed kernels changes: sha256 over the product's HIP sources and the headers
    they include, comments and white space removed, + the compiler flags.  (The sources, not the built library: the .so is rebuilt
    wherever the tree is checked out, the sources are what is committed.)"""
    import hashlib
    import re
    h = hashlib.sha256()
    base = os.path.join(ROOT, "vortex-raytracing_amd", "csrc")
    for f in ("rt_kernels.hip", "rt_types.h", os.path.join("..", "..", "include", "vortex_hip.h")):
        try:
            t = open(os.path.join(base, f)).read()
        except OSError:
            return None
        t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
        t = re.sub(r"//[^\n]*", " ", t)
        h.update(re.sub(r"\s+", " ", t).encode())
    try:
        bld = importlib.import_module("vortex-raytracing_amd.build")
        h.update(" ".join(bld.HIP_FLAGS).encode())
    except Exception:
        return None
    return h.hexdigest()[:16]


def tree_id(scene):
    """16 hex digits of the tree the frame is traced through: sha256 over the reference-format TLAS, instance, BVH and triangle
    buffers (the compact layout the kernels read is a deterministic function of them)."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for k in ("tlas", "blas", "bvh", "tri"):
        h.update(np.ascontiguousarray(scene[k]).view(np.uint8).tobytes())
    return h.hexdigest()[:16]


def profile_mismatch(prof, ident):
    """Why the checked-in per-frame counters (profiles/valu_profile.json) do NOT describe this run, or None if they do: the file
    carries the identity of the run it was taken from -- frame size, tree hash and node count, kernel source id, and the traversal
    counts the counting build reported there.  All must equal this run's, except the two fetch counts, which may differ by 0.1 %: the counting
    build's totals move by a few thousandths of a per cent between runs (which of a tile's rays a lane shares its leaf pass with depends on the
    order tiles were handed out), far below what a different tree, frame or kernel would change them by."""
    want = prof.get("identity")
    if not want:
        return "profiles/valu_profile.json carries no identity block (written by an older tools/roofline_from_pmc.py)"
    for k in ("width", "height", "shadow", "bvh_nodes", "tree_sha16", "kernel_source_sha16", "node_fetches_timed", "tri_fetches_timed", "rays"):
        a, b = want.get(k), ident.get(k)
        if k.endswith("_fetches_timed") and a and b and abs(a - b) <= 1e-3 * max(a, b):
            continue
        if a != b:
            return "profiles/valu_profile.json was taken on another %s (%r there, %r here): re-run tools/round_profile.sh" % (k, want.get(k), ident.get(k))
    return None


def other_configs(vrt, torch, dev, ds, scene, params, full):
    """BASELINE.json's other configurations as short legs of this run (they are parity-test cases -- tests/test_gpu_configs.py -- not the bench
    line): strictly serial frames through the C ABI, inputs resident, each leg timed by the host clock between two device
    synchronisations, rays counted by the launches themselves in an untimed frame.  Runs in a child process of the bench (see main);
    tools/config_bench.py holds the longer versions."""
    rtapi = vrt.rtapi
    s = torch.cuda.current_stream().cuda_stream
    rtapi.accel_frames_in_flight(ds.accel, 1)

    def leg(name, first, frame, frames, **more):
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        first(cnt.data_ptr())
        torch.cuda.synchronize()
        rays = int(cnt.item())
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        assert rtapi.status(s) == 0
        return dict({"config": name, "frames_timed": frames, "rays_per_frame": rays, "ms_per_frame": round(ms, 4), "mrays_s": round(rays / ms / 1e3, 1)}, **more)

    out = []
    W, H = 1920, 1080
    px = torch.zeros((2160, 3840), dtype=torch.int32, device=dev)
    out.append(leg("configs[2] as worded: Sponza-class, 1920x1080, primary + 1 diffuse bounce (vxrt_render_diffuse_bounce)",
                   lambda c: rtapi.render_diffuse_bounce(ds.accel, W, H, 0, H, params, px.data_ptr(), seed=3, rays_ptr=c, stream=s),
                   lambda: rtapi.render_diffuse_bounce(ds.accel, W, H, 0, H, params, px.data_ptr(), seed=3, stream=s), 40, tris=scene.n_tris))
    out.append(leg("configs[3] on one GPU: Sponza-class, 3840x2160, primary + 1 shadow ray",
                   lambda c: rtapi.render(ds.accel, 3840, 2160, 0, 2160, params, px.data_ptr(), 1, None, None, c, s),
                   lambda: rtapi.render(ds.accel, 3840, 2160, 0, 2160, params, px.data_ptr(), 1, None, None, None, s), 20, tris=scene.n_tris))
    t0 = time.perf_counter()
    bunny = vrt.scene.procedural("bunny", 6, 0, 1)
    bs = time.perf_counter() - t0
    db = vrt.tracer.DeviceScene(bunny, dev)
    pb = rtapi.default_shade_params()
    pb.light_pos[:] = (20.0, 260.0, -150.0)
    out.append(leg("configs[1]: bunny-class (framed), 1024x1024, primary + 1 shadow ray",
                   lambda c: rtapi.render(db.accel, 1024, 1024, 0, 1024, pb, px.data_ptr(), 1, None, None, c, s),
                   lambda: rtapi.render(db.accel, 1024, 1024, 0, 1024, pb, px.data_ptr(), 1, None, None, None, s), 100, tris=bunny.n_tris, host_build_s=round(bs, 2)))
    db.close()
    if full:
        import numpy as np
        t0 = time.perf_counter()
        hair = vrt.scene.procedural("hairball_fill", 20000, 250, 7)
        bs = time.perf_counter() - t0
        dh = vrt.tracer.DeviceScene(hair, dev)
        b = hair.bounds
        radius = 0.25 * 0.5 * float(np.linalg.norm(np.array(b[3:]) - np.array(b[:3])))
        ph = rtapi.default_shade_params()
        ph.light_pos[:] = (0.0, 400.0, 0.0)
        out.append(leg("configs[4]: hairball (framed), 1920x1080, 16 spp AO (tmax = 0.25 scene radius)",
                       lambda c: rtapi.render_ao(dh.accel, W, H, 0, H, ph, 16, radius, px.data_ptr(), seed=7, rays_ptr=c, stream=s),
                       lambda: rtapi.render_ao(dh.accel, W, H, 0, H, ph, 16, radius, px.data_ptr(), seed=7, stream=s), 3, tris=hair.n_tris, host_build_s=round(bs, 1)))
        dh.close()
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here, as fresh child processes (one per
    GPU, `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`), BEFORE anything in this process touches
    HIP (the device count itself is taken in a child process), relay rank 0's JSON line and exit with the children's code.
    Never exec: the parent stays a plain process.  Fewer than N visible devices is an error, not an N=1 run."""
    import socket
    import subprocess
    # the device count comes from a child process: nothing in THIS process may open the HIP runtime before the ranks are started
    # (torch.cuda.device_count() can fall through to hipGetDeviceCount on builds without amdsmi)
    try:
        have = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True, timeout=300).stdout.strip().splitlines()[-1])
    except Exception:
        have = 0
    if a.dist_backend == "nccl" and have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to report an N=%d run as N=%d\n" % (a.gpus, have, have, a.gpus))
        return 3
    if have < 1:
        sys.stderr.write("bench.py needs a HIP device: there is no CPU fallback for the hot path\n")
        return 3
    with socket.socket() as so:      # a free rendezvous port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    if r.returncode == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: the %d ranks printed %d JSON lines, expected exactly one (rank 0)\n" % (a.gpus, len(lines)))
        return 4
    for l in lines:
        print(l, flush=True)
    return r.returncode


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))
    # ONE line on stdout: whatever native libraries print there (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    if a.other_configs_child:
        # the legs of the other configurations, in a process of their own (started by rank 0 of a 1-GPU run once its own figures are
        # taken): a process with the bench's streams alive maps more streams than the card has hardware queues, and a small frame's main
        # and side stream then share one (the bunny-class frame measured 0.50 ms there, 0.35 ms alone)
        torch.cuda.set_device(0)
        vrt = importlib.import_module("vortex-raytracing_amd")
        scene = vrt.scene.procedural("atrium", a.level, 0, 3)
        ds = vrt.tracer.DeviceScene(scene, "cuda:0")
        params = vrt.rtapi.default_shade_params()
        params.light_pos[:] = (300.0, 480.0, 60.0)
        legs = other_configs(vrt, torch, "cuda:0", ds, scene, params, a.other_configs_child == "full")
        os.write(json_fd, (json.dumps(legs) + "\n").encode())
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # a launcher started a different number of ranks than --gpus names: the line would claim the wrong N
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (a.gpus, world))
    # rehearsal of rank 0 of an N-GPU run on one GPU, without the network: everything a rank does per frame except the collective
    rehearse = a.rehearse_world if (world == 1 and a.rehearse_world > 1) else 0
    if rehearse:
        world = rehearse
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback for the hot path")
    if a.dist_backend == "gloo":
        local %= max(1, torch.cuda.device_count())   # rehearsal: ranks may share a GPU
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (LOCAL_RANK %d, %d visible): one GPU per rank with the nccl backend" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    multi = world > 1 or a.one_rank_group          # take the N-rank code path (shares, batches, image assembly)
    grouped = multi and not rehearse                # ... with a process group and its collectives
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if a.dist_backend == "nccl" else "cpu"   # where collectives' tensors live
    # what the line reports about the job: the group's own world size and the device each rank runs on
    my_dev = "%s (%s)" % (dev, torch.cuda.get_device_name(local))
    rank_devices = [my_dev]
    dist_world = 1
    if grouped:
        dist_world = dist.get_world_size()
        rank_devices = [None] * dist_world
        dist.all_gather_object(rank_devices, my_dev)

    vrt = importlib.import_module("vortex-raytracing_amd")
    rtapi, sharding = vrt.rtapi, vrt.sharding

    W, H = a.width, a.height
    scene = vrt.scene.procedural("atrium", a.level, 0, 3)
    ds = vrt.tracer.DeviceScene(scene, dev)
    shadow = 0 if a.no_shadow else 1
    params = rtapi.default_shade_params()
    LIGHT = (300.0, 480.0, 60.0)
    params.light_pos[:] = LIGHT   # inside the hall, so shadow rays are real work

    y0, y1 = sharding.row_bands(H, world)[rank] if (a.shard == "rows" and world > 1) else (0, H)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    nfl = max(1, min(8, a.frames_in_flight))
    # frame i goes to stream i % nfl and framebuffer i % nfl; the accel keeps nfl frame contexts
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(nfl - 1)]
    if os.environ.get("VXRT_BENCH_OWN_STREAMS"):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    # (with several ranks a framebuffer stays busy until its gather has run: twice as many, so that rendering never waits for the link)
    n_frames = max(2, nfl) * (2 if multi else 1)
    ig = bg = None
    bands = multi and a.shard == "bands"
    band_plan = None
    # Several ranks: a rank's share of one frame is small against its GPU (4,080 tiles for 8,192 resident wavefronts at 1080p / 8)
    # and takes as long as its slowest tile -- about half a full frame's time, whatever N.  The steps are therefore issued in sets
    # of B frames per set of launches and assembled with one collective per set.
    B = 1
    set_sizes = None            # bands: sizes of the timed region's sets
    if bands:
        B = max(1, min(32, a.batch if a.batch > 0 else 16))
        if a.taper == "auto" and a.batch <= 0:
            set_sizes = sharding.taper(a.steps, B, 1, 0.5)
        elif a.taper not in ("auto", "none"):
            set_sizes = [int(x) for x in a.taper.split(",")]
            if sum(set_sizes) != a.steps or min(set_sizes) < 1 or max(set_sizes) > 32:
                raise SystemExit("bench.py: --taper must list set sizes of 1..32 frames that add up to --steps")
        else:
            n_groups = max(nfl, -(-a.steps // B))
            per = -(-a.steps // n_groups)
            set_sizes = [min(per, a.steps - g) for g in range(0, a.steps, per)]
        B = max(set_sizes)
        frames = None           # allocated once the band plan is known (below)
    elif multi and a.shard == "tilerows":
        # default: as large as a batch may be (16), but the K timed steps split into equal groups, at least one per stream
        # (K = 20 -> 2 groups of 10, not 16 + 4)
        n_groups = max(nfl, -(-a.steps // 16))
        B = max(1, min(32, a.batch if a.batch > 0 else -(-a.steps // n_groups)))
        if a.sets != "auto":
            set_sizes = [int(x) for x in a.sets.split(",")]
            if sum(set_sizes) != a.steps or min(set_sizes) < 1 or max(set_sizes) > 32:
                raise SystemExit("bench.py: --sets must list set sizes of 1..32 frames that add up to --steps")
            B = max(set_sizes)
        elif a.batch <= 0 and a.steps >= 2 * world and world > 1:
            # a last set of N frames (one frame's bytes per link whatever N is), the steps before it in equal sets of at most 16:
            # sharding.auto_sets; what each schedule costs rank 0's pipeline is in profiles/r05_e_multi_gpu_schedules.txt
            set_sizes = sharding.auto_sets(a.steps, world)
            B = max(set_sizes)
        ig = sharding.InterleavedGather(H, W, rank, world, dev, slots=n_frames, collective=not rehearse, batch=B, single_rank_collective=a.one_rank_group, wire=a.wire)
        frames = [ig.new_frame_buffer(dev) for _ in range(n_frames)]
    elif not multi and a.batch != 1 and nfl > 1:
        # One GPU: whole frames in groups of up to 5 per set of launches (the same entry point with one rank).  Each wavefront then
        # works through 5x as many tiles per launch, so ramp and tail of a launch weigh less -- the effect that lets a 3840x2160
        # frame reach 10 Grays/s: +5..6 % (3..12 frames per set measured alike: profiles/r02_l_frame_batches.txt).
        # --batch 1 = one frame per set of launches.
        n_groups = max(nfl, -(-a.steps // 5))
        B = max(1, min(32, a.batch if a.batch > 1 else -(-a.steps // n_groups)))
        frames = [torch.zeros((B, H, W), dtype=torch.int32, device=dev) for _ in range(n_frames)]
    else:
        frames = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(n_frames)]
    counters = torch.zeros(8, dtype=torch.int64, device=dev)
    frame_stride = ig.frame_stride if ig is not None else H * W

    if bands:
        # ---- the band plan: cut the rows where the COST is equal.  Every rank times sets of frames of its own band (GPU events, nothing
        # else running), the times are all-gathered, every rank computes the same new cut from them (sharding.rebalance_bands); a
        # few rounds, the plan with the smallest slowest-rank time is kept.  All of it before the warmup steps, none of it timed.
        # (The rehearsal on one GPU times every rank's band in turn instead of gathering.)
        bounds = sharding.equal_bands(H, world)
        cal_k = 4
        cal = torch.zeros((cal_k, H, W), dtype=torch.int32, device=dev)

        def band_ms(r, bnds, reps=3):
            b0, b1 = bnds[r], bnds[r + 1]
            best = None
            for i in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rtapi.render_rows_batch(ds.accel, W, H, b0, b1, [params] * cal_k, cal.data_ptr(), H * W, shadow, None, sptr)
                e1.record(stream)
                torch.cuda.synchronize()
                if i:                                   # (the first call of a new window builds its tables)
                    t = e0.elapsed_time(e1) / cal_k
                    best = t if best is None else min(best, t)
            return best

        history = []
        for it in range(5 if world > 1 else 0):
            if rehearse:
                times = [band_ms(r, bounds) for r in range(world)]
            else:
                mine = torch.tensor([band_ms(rank, bounds)], dtype=torch.float64, device=cdev)
                allt = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allt, mine)
                times = [float(t.item()) for t in allt]
            history.append((max(times), list(bounds), [round(t, 4) for t in times]))
            nb = sharding.rebalance_bands(bounds, times, H, damping=1.0 if it < 2 else 0.5)
            if nb == bounds or any(nb == h[1] for h in history):
                break
            bounds = nb
        if history:
            best = min(history, key=lambda h: h[0])
            bounds = best[1]
            band_plan = {"bounds": bounds, "ms_per_frame_by_rank_when_planned": best[2], "rounds": len(history),
                         "imbalance_max_over_mean": round(best[0] / (sum(best[2]) / len(best[2])), 4)}
        del cal
        y0, y1 = bounds[rank], bounds[rank + 1]
        n_slots = min(len(set_sizes) + 2, 4)
        slot_sizes = [B] * n_slots
        bg = sharding.BandGather(H, W, rank, world, bounds, dev, slot_sizes, collective=not rehearse, via_cpu=(cdev == "cpu"))
        frames = bg.bufs

    def launch(buf, count_ptr=None, st=None, k=1, slot=0):
        sp = (st or stream).cuda_stream
        if bands:
            dst, fstride = bg.target(slot)
            if count_ptr is None:
                rtapi.render_rows_batch(ds.accel, W, H, y0, y1, [params] * k, dst, fstride, shadow, None, sp)
            else:
                rtapi.render(ds.accel, W, H, y0, y1, params, dst, shadow, None, None, count_ptr, sp)
        elif B > 1 and k > 0 and count_ptr is None and st is not None:
            if not multi:
                rtapi.render_batch(ds.accel, W, H, [params] * k, buf.data_ptr(), frame_stride, shadow, None, sp)
            else:
                rtapi.render_interleaved_batch(ds.accel, W, H, rank, world, [params] * k, buf.data_ptr(), frame_stride, shadow, None, sp)
        elif multi and a.shard == "tilerows":
            rtapi.render_interleaved(ds.accel, W, H, rank, world, params, buf.data_ptr(), shadow, None, None, count_ptr, sp)
        else:
            if count_ptr is None and os.environ.get("VXRT_BENCH_COUNT_RAYS"):      # (diagnostic: what the optional rays counter costs a frame)
                count_ptr = counters.data_ptr()
            rtapi.render(ds.accel, W, H, y0, y1, params, buf.data_ptr(), shadow, None, None, count_ptr, sp)

    # rays per step on this rank (primary + shadow), counted once by the kernel itself
    launch(frames[0], counters.data_ptr())
    torch.cuda.synchronize()
    assert rtapi.status(sptr) == 0, "kernel status (traversal stack overflow)"
    rays_rank = int(counters[0].item())
    algo = algo_timed = None
    if world == 1:
        algo = rtapi.render_stats(ds.accel, W, H, 0, H, params, frames[0].data_ptr(), shadow, sptr)                  # reference-order counts
        algo_timed = rtapi.render_stats(ds.accel, W, H, 0, H, params, frames[0].data_ptr(), shadow, sptr, timed=True)  # the traversal that is timed
    elif rank == 0:
        # several ranks: rank 0 counts the WHOLE frame once (into a scratch frame), so that the N-rank line can say which tree and
        # which traversal its roofline numerator belongs to; the timed region traces the ranks' bands only
        scratch = torch.zeros((H, W), dtype=torch.int32, device=dev)
        algo_timed = rtapi.render_stats(ds.accel, W, H, 0, H, params, scratch.data_ptr(), shadow, sptr, timed=True)
        del scratch

    rtapi.accel_frames_in_flight(ds.accel, nfl)
    gather_stream = torch.cuda.Stream(device=dev) if multi else None
    gdone = [None] * len(frames)   # per framebuffer: event of the last gather that read it

    no_gather = bool(rehearse and os.environ.get("VXRT_BENCH_NO_GATHER"))   # (diagnostic: the share's launches alone)

    def step(i, ev=None, k=1):
        """Issues launch group i: k steps (k = 1 except with several ranks, where a group is a set of up to B frames)."""
        b = i % len(frames)
        buf = frames[b]
        st = streams[i % nfl]
        if gdone[b] is not None:
            st.wait_event(gdone[b])            # do not overwrite a frame that is still being gathered
        if ev is not None:
            ev[0].record(st)
        launch(buf, st=st, k=k, slot=b)
        if ev is not None:
            ev[1].record(st)
        if not (multi and not no_gather):
            return None
        done = torch.cuda.Event()
        done.record(st)

        def assemble():
            # image assembly overlaps the next set's traversal: the gather runs on its own stream, behind this set's launches
            gather_stream.wait_event(done)
            with torch.cuda.stream(gather_stream):
                if bands:
                    bg.gather(b, k)
                elif a.shard == "tilerows":
                    ig.gather(buf, b, via_cpu=(cdev == "cpu"), k=(k if B > 1 else None))
                else:
                    band = buf[y0:y1]
                    sharding.gather_frame(band if cdev != "cpu" else band.cpu(), H, W, rank, world)
                gdone[b] = torch.cuda.Event()
                gdone[b].record(gather_stream)
        return assemble

    def issue(seq, evs_=None, marks=None):
        """Issues the launch groups seq = [(i, k), ...] in order.  A group's image assembly (packing, collective, unpacking: a dozen calls
        on the host) is enqueued only after the NEXT group's launches are out (with nfl streams: after the next nfl - 1 groups'), so that
        the sets that are to run side by side are on the device within microseconds of each other instead of a host call sequence apart
        (the first set used to have the machine's half to itself for 0.16 ms of a 1.2 ms run)."""
        pend = []
        for n, (i, k) in enumerate(seq):
            fin = step(i, evs_[n] if evs_ else None, k=k)
            if fin is not None:
                pend.append(fin)
            if marks is not None:
                marks.append(time.perf_counter())
            while len(pend) >= max(1, nfl):
                pend.pop(0)()
        for fin in pend:
            fin()

    # isolated duration of one step's launches (nothing else on the GPU): HIP events on the launch stream
    iso = []
    for i in range(min(10, max(3, a.steps))):
        e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e[0].record(stream)
        launch(frames[0])
        e[1].record(stream)
        torch.cuda.synchronize()
        iso.append(e[0].elapsed_time(e[1]))
    iso_ms = sum(iso) / len(iso)
    def groups(n):     # n steps as launch groups of at most B
        return [min(B, n - g) for g in range(0, n, B)]
    timed_groups = list(set_sizes) if set_sizes else groups(a.steps)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in timed_groups]
    # The GPU raises its clocks over the first tens of milliseconds of sustained load (measured: the frame period of a 20-step
    # run shrinks from 0.52 to 0.49 ms between its first and last step).  A short run would time that ramp, not the path, so the
    # clocks are brought to their sustained state first with untimed frames of the same kind; then the W warmup steps, which run as
    # the timed ones do (same streams, same frames in flight) and directly before them.
    settled = a.settle_frames
    # (every stream of the timed region carries work before the region opens: a stream's hardware queue is created at its first use, 5 ms
    # that a run whose schedule leaves a stream idle until the closing wait_stream would otherwise time)
    for st_ in streams + ([gather_stream] if multi else []):
        torch.cuda.Event().record(st_)
    if set_sizes:
        # the settle phase repeats the timed region's own sequence of sets on the same streams and buffers (a frame context learns
        # its longest-tile-first order per set size from its previous set of that size), then the W warmup steps as a prefix of it
        settled = 0
        while settled < a.settle_frames:
            issue(list(enumerate(timed_groups)))
            settled += sum(timed_groups)
        left, wseq = a.warmup, []
        for j, k in enumerate(timed_groups):
            if left <= 0:
                break
            wseq.append((j, min(k, left)))
            left -= min(k, left)
        issue(wseq)
    else:
        issue(list(enumerate(groups(a.settle_frames) + groups(a.warmup))))
    # every frame rendered before the timed region opens (one for the ray count, two counting builds, the isolated launches,
    # the band-planning sets, the clock-settling frames, the W warmup steps): none of them is timed, none of their results is reused
    frames_before = 1 + (2 if world == 1 else (1 if rank == 0 else 0)) + len(iso) + settled + a.warmup + (band_plan["rounds"] * 16 * (world if rehearse else 1) if band_plan else 0)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    probe = ClockProbe(vrt, torch, dev) if rank == 0 else None
    if probe:
        probe.launch(0, sptr)          # (untimed: behind the warmup steps, before the region's opening synchronisation)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_marks = []
    issue(list(enumerate(timed_groups)), evs, host_marks)
    host_marks = [m - t0 for m in host_marks]
    # end of the timed region on the GPU's clock: an event on the first stream behind every stream of the region
    for st_ in streams[1:] + ([gather_stream] if multi else []):
        stream.wait_stream(st_)
    end_ev = torch.cuda.Event(enable_timing=True)
    end_ev.record(stream)
    t_issued = time.perf_counter() - t0     # host time to issue the K steps (diagnostic: a host-bound run has t_issued ~ elapsed)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if probe:
        # the second clock probe, right behind the region's closing synchronisation (tens of microseconds after the last step: the chip moves its
        # clock over milliseconds).  A measurement's own kernel -- a 30 us spin -- is not part of the K steps and is not timed with them.
        probe.launch(1, sptr)
        torch.cuda.synchronize()
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_rank], dtype=torch.int64, device=cdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_all = int(r.item())        # the rays of ONE frame: every rank traces a disjoint part of it
    else:
        rays_all = rays_rank
    assert rtapi.status(sptr) == 0
    # HIP events on the launch streams.  With frames in flight the launches of consecutive steps overlap, so the per-launch
    # duration that prices the roofline is the span of the timed region's events divided by the launches in it (their union,
    # not their sum); the overlapped and the isolated per-launch durations are reported next to it.
    ovl_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / a.steps
    span_ms = evs[0][0].elapsed_time(end_ev)
    kern_ms = span_ms / a.steps
    if os.environ.get("VXRT_BENCH_TRACE") and rank == 0:   # start offset of every timed step on the GPU's clock (debugging the timed region itself)
        print("step starts (ms):", " ".join("%.3f" % evs[0][0].elapsed_time(e0) for e0, _ in evs), "end %.3f" % span_ms, "host %.3f" % (elapsed * 1e3), "issued %.3f" % (t_issued * 1e3),
              "host time after each step's calls returned (ms): " + " ".join("%.3f" % (m * 1e3) for m in host_marks), file=sys.stderr)

    extras = {}
    random_sample = None
    if a.random_rays:
        # north_star's second figure: synthetic random rays against the fixed BVH, ray buffer (24 B/ray) in HBM ->
        # hit records (24 B/ray); every rank traces its own N rays (seed 12345 + rank), no collective
        n = a.random_rays
        g = torch.Generator(device=dev).manual_seed(12345 + rank)
        lo = torch.tensor(scene.bounds[:3], device=dev)
        hi = torch.tensor(scene.bounds[3:], device=dev)
        o = lo + (hi - lo) * torch.rand((n, 3), generator=g, device=dev)
        d = torch.randn((n, 3), generator=g, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        rays = torch.cat([o, d], 1).contiguous()
        del o, d
        random_sample = rays[:60000].cpu().numpy() if rank == 0 else None     # (for the CPU baseline's like-for-like leg)
        hits = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
        rstats = rtapi.trace_stats(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr) if rank == 0 else None
        reps = 5
        for _ in range(2):
            rtapi.trace(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            rtapi.trace(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        rt = time.perf_counter() - t1
        if grouped:
            t = torch.tensor([rt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rt = float(t.item())
        assert rtapi.status(sptr) == 0
        mr = n * world * reps / rt / 1e6
        extras["random_rays_mrays_s"] = round(mr, 1)
        extras["random_rays_n"] = n * world
        extras["random_rays_scaling"] = "weak (N rays per GPU)"
        if rstats:
            gbs = mr * 1e6 * rstats["bytes_per_ray"] / 1e9
            extras["random_rays"] = {"rays_per_gpu": n, "mrays_s": round(mr, 1), "ms_per_launch": round(rt / reps * 1e3, 3),
                                     "bytes_per_ray": round(rstats["bytes_per_ray"], 1), "algorithmic_GBs": round(gbs, 1),
                                     "frac_of_hbm_peak": round(gbs / (HBM_PEAK_GBS * world), 4), "frac_of_l2_peak": round(gbs / (L2_PEAK_GBS * world), 4),
                                     "node_fetches_per_ray": round(rstats["node_fetches"] / n, 2),
                                     "tri_fetches_per_ray": round(rstats["tri_fetches"] / n, 2),
                                     "node_fetches": rstats["node_fetches"], "tri_fetches": rstats["tri_fetches"]}
        del rays, hits

    if multi and a.shard == "tilerows" and a.leg_4k and (W, H, a.level) == (1920, 1080, 8) and B > 1:
        # BASELINE configs[3], the workload it names for 8 GPUs: the same scene at 3840x2160, the framebuffer's tile rows interleaved over
        # the ranks, one gather per set -- a second leg of the N > 1 line, through the very machinery of the first (the closures above see
        # the rebound frame size, buffers and assembly).  2 N timed frames as two sets, the last one N / 4 frames (one 1080p frame's bytes
        # per link), behind a settle phase of its own; value = the whole frame's rays x frames / max-over-ranks time.
        main_leg = (W, H, ig, frames, gdone, B, frame_stride)
        try:
            W, H = 3840, 2160
            last4 = max(1, world // 4)              # a 3840x2160 frame is four 1080p frames: N / 4 of them put one 1080p frame's bytes on each link
            B = max(1, min(16, 2 * world - last4))
            ig = sharding.InterleavedGather(H, W, rank, world, dev, slots=4, collective=not rehearse, batch=B, single_rank_collective=a.one_rank_group, wire=a.wire)
            frames = [ig.new_frame_buffer(dev) for _ in range(4)]
            gdone = [None] * len(frames)
            frame_stride = ig.frame_stride
            cnt4 = torch.zeros(1, dtype=torch.int64, device=dev)
            rtapi.render_interleaved(ds.accel, W, H, rank, world, params, frames[0].data_ptr(), shadow, None, None, cnt4.data_ptr(), sptr)
            torch.cuda.synchronize()
            rays4 = int(cnt4.item())
            seq = [(0, B), (1, last4)] if B > 1 else [(0, 1), (1, 1)]
            for _ in range(4):                      # settle: the leg's own sequence (tile orders are learned per set size and window)
                issue(seq)
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            issue(seq)
            for st_ in streams[1:] + [gather_stream]:
                stream.wait_stream(st_)
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()
            el4 = time.perf_counter() - t4
            if grouped:
                t = torch.tensor([el4], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el4 = float(t.item())
                r = torch.tensor([rays4], dtype=torch.int64, device=cdev)
                dist.all_reduce(r, op=dist.ReduceOp.SUM)
                rays4 = int(r.item())
            assert rtapi.status(sptr) == 0
            extras["leg_3840x2160"] = {"config": "configs[3]: Sponza-class, 3840x2160, primary + 1 shadow ray, tile rows interleaved over %d ranks" % world,
                                       "steps": sum(k for _, k in seq), "sets": [k for _, k in seq], "ms_per_step": round(el4 / sum(k for _, k in seq) * 1e3, 4),
                                       "rays_per_step" + ("_rank0" if rehearse else ""): rays4,
                                       "mrays_s" + ("_rank0_share_only" if rehearse else ""): round(rays4 * sum(k for _, k in seq) / el4 / 1e6, 1),
                                       "wire_bytes_per_rank_last_set": ig.wire_bytes(seq[-1][1]), "wire_format": ig.wire}
        except Exception as e:      # (a failed side leg never takes the bench line with it)
            extras["leg_3840x2160"] = {"error": repr(e)[:300]}
        finally:
            W, H, ig, frames, gdone, B, frame_stride = main_leg

    if world == 1 and a.other_configs != "none" and (W, H, a.level) == (1920, 1080, 8):
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--other-configs-child", a.other_configs], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("[")]
            extras["other_configs"] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:      # (a failed side leg never takes the bench line with it)
            extras["other_configs"] = {"error": repr(e)[:300]}

    if rank == 0:
        if not multi:
            par = "1 GPU: whole frames, %d per set of launches (vxrt_render_batch), %d sets in flight" % (B, nfl) if B > 1 else "1 GPU: whole frame"
        elif bands:
            par = ("one frame split into %d contiguous bands of rows cut at equal measured cost on 8-row tile boundaries (rank r: rows bounds[r]..bounds[r+1]), "
                   "frames traced in sets of %s per set of launches, each band received in place on rank 0: one group of RCCL send/recv (a gather with per-rank sizes) per set"
                   % (world, "/".join(str(k) for k in timed_groups)))
        elif a.shard == "tilerows":
            par = "one frame split by interleaved 8-row tile rows (rank r: rows r, r+%d, ... of %d tile rows) x%d GPUs, %d frames per set of launches, one RCCL gather of the shares to rank 0 per set" % (world, (H + 7) // 8, world, B)
        else:
            par = "one frame split into %d contiguous tile-aligned row bands, RCCL gather to rank 0" % world
        out = {
            "metric": "Mrays/s (primary+shadow) at %dx%d, 1M-tri BVH" % (W, H),
            "value": round(rays_all * a.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",   # the frame -- the total work -- is fixed as N grows
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (procedural 'Sponza-class' atrium, seed 3; no Sponza/bunny asset exists offline)",
            "config": {"workload": "configs[2]: Sponza-class %d tris, %dx%d, primary + 1 shadow ray per hit%s" % (scene.n_tris, W, H, "" if shadow else " (shadow disabled)"),
                       "rays_per_step": rays_all, "rays_per_step_rank0": rays_rank, "frames_in_flight": nfl, "frames_per_launch_group": B, "clock_settle_frames_untimed": settled, "frames_rendered_before_the_timed_region": frames_before, "parallelism": par,
                       "world_size": dist_world, "rank_devices": rank_devices, "dist_backend": (a.dist_backend if grouped else None),
                       "bvh_nodes": scene.n_bvh_nodes, "bvh_depth": scene.info.get("max_depth")},
        }
        if bands:
            out["config"].update({"shard": "bands", "sets_of_the_timed_steps": timed_groups, "band_plan": band_plan, "rows_rank0": [y0, y1]})
        elif multi:
            out["config"]["shard"] = a.shard
            if ig is not None:
                out["config"].update({"sets_of_the_timed_steps": timed_groups, "wire_format": ig.wire,
                                      "wire_bytes_per_rank_last_set": ig.wire_bytes(timed_groups[-1])})
        prof = load_profile_constants()
        # one step = the launches of vxrt_render: persistent traversal kernel (dominant, > 93 % of the step), the EXACT launches
        # for the rays with NaN-capable slabs, and the shading pass; priced together
        clk = [probe.ghz(0), probe.ghz(1)] if probe else [None, None]
        clock_held = (sum(clk) / 2.0) if all(clk) else None
        clock = clock_held or CLOCK_GHZ
        # ONE denominator: wave64 VALU instructions per second against what 1,024 SIMDs can issue at the clock held in the timed
        # region, at the guide's 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md: "issues each VALU instruction over 2
        # cycles").  The same fraction at the 2.2 cycles this chip measured for dense independent streams (tools/calibrate_valu.py)
        # and at the nominal 2.4 GHz are given next to it, labelled.
        roof = {"bound": "valu", "achieved": None, "peak": round(SIMDS * clock / 2.0, 1), "unit": "G wave64 VALU instructions/s", "frac": None, "traffic": None,
                "peak_is": "1024 SIMDs x clock held in the timed region / 2 cycles per wave64 VALU instruction (guide constant)",
                "clock_ghz_held": round(clock_held, 4) if clock_held else None,
                "clock_probe_ghz_before_after": [round(c, 4) if c else None for c in clk],
                "clock_source": "one-wavefront s_memtime / s_memrealtime probe right before the timed region opens and right after it closes (lib/libvxrt_calib.so)" if clock_held
                                else "nominal %.1f GHz (probe library not available)" % CLOCK_GHZ,
                "kernel": "rt_persistent_kernel<JOB_RENDER%s> (+ EXACT launches + rt_shade_kernel)" % ("_SHADOW" if shadow else ""),
                "kernel_ms": round(kern_ms, 4), "kernel_ms_overlapped": round(ovl_ms, 4), "kernel_ms_isolated": round(iso_ms, 4),
                "kernel_ms_is": "GPU-clock span of the timed region (HIP events: first step's start -> an event behind every stream) / steps",
                "frames_per_launch": B, "launch_set_ms_overlapped": round(ovl_ms * a.steps / max(1, len(evs)), 4),
                "frames_in_flight": nfl,
                "why_valu": "the scene is cache-resident (measured HBM traffic = a few % of the HBM peak) and relieving the memory path measured neutral (LDS-staged "
                            "top of the tree: 39 % of node steps from LDS, -17 % vector-memory instructions, +0 %: profiles/r02_b_lds_top_counters.txt); what moves the time "
                            "is the number of VALU instructions"}
        # what the per-frame profile constants must have been taken on to be usable here (and what a new profile pass records)
        ident = {"width": W, "height": H, "shadow": int(shadow), "bvh_nodes": scene.n_bvh_nodes, "tree_sha16": tree_id(scene),
                 "kernel_source_sha16": kernel_source_id(),
                 "node_fetches_timed": algo_timed["node_fetches"] if algo_timed else None,
                 "tri_fetches_timed": algo_timed["tri_fetches"] if algo_timed else None,
                 "rays": algo_timed["rays"] if algo_timed else None}
        roof["identity"] = ident
        why_not = None
        if not prof:
            why_not = "profiles/valu_profile.json not found"
        else:
            why_not = profile_mismatch(prof, ident)
        if why_not:
            roof["frac_is_null_because"] = why_not
        if prof and world > 1 and not why_not and not rehearse:
            # N ranks: the frame's instructions are a property of the frame, whoever traces which band: the whole job's rate against
            # N x the per-GPU roof (clock: rank 0's probe)
            n_valu = prof["valu_instr_per_frame"]
            ach = n_valu / (elapsed / a.steps) / 1e9
            roof.update({"achieved": round(ach, 1), "peak": round(world * SIMDS * clock / 2.0, 1), "frac": round(ach / (world * SIMDS * clock / 2.0), 4),
                         "peak_is": "%d GPUs x 1024 SIMDs x clock held on rank 0 / 2 cycles per wave64 VALU instruction" % world,
                         "achieved_is": "wave64 VALU instructions of one whole frame (profile constant) x steps / max-over-ranks wall time of the timed region",
                         "valu_instr_per_frame": n_valu, "valu_source": prof.get("source")})
        if prof and world == 1 and not why_not:
            # profile constants of THIS workload (instruction counts per frame are a property of the frame and the code), source named in the file
            n_valu = prof["valu_instr_per_frame"]
            ach = n_valu / (kern_ms * 1e-3) / 1e9
            mk = prof.get("main_kernel", {})
            roof.update({"achieved": round(ach, 1), "frac": round(ach / (SIMDS * clock / 2.0), 4),
                         "frac_isolated": round(n_valu / (iso_ms * 1e-3) / 1e9 / (SIMDS * clock / 2.0), 4),
                         "valu_instr_per_frame": n_valu, "valu_source": prof.get("source"),
                         "same_numerator_other_denominators": {
                             "frac_at_2.2_cycles_per_instruction_measured_on_this_chip": round(ach / (SIMDS * clock / 2.2), 4),
                             "frac_at_nominal_2.4_GHz_and_2_cycles": round(ach / (SIMDS * CLOCK_GHZ / 2.0), 4),
                             "frac_at_nominal_2.4_GHz_and_2.2_cycles": round(ach / (SIMDS * CLOCK_GHZ / 2.2), 4)}})
            if mk.get("SQ_ACTIVE_INST_VALU") and mk.get("SQ_WAVE_CYCLES"):
                roof["main_kernel_counters"] = {
                    "lane_utilisation": round(mk["SQ_THREAD_CYCLES_VALU"] / (64.0 * mk["SQ_ACTIVE_INST_VALU"]), 4),
                    "lane_utilisation_is": "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
                    "wait_any_of_wave_cycles": round(mk["SQ_WAIT_ANY"] / mk["SQ_WAVE_CYCLES"], 4),
                    "wait_inst_any_of_wave_cycles": round(mk["SQ_WAIT_INST_ANY"] / mk["SQ_WAVE_CYCLES"], 4),
                    "salu_per_valu": round(mk["SQ_INSTS_SALU"] / mk["SQ_INSTS_VALU"], 4) if mk.get("SQ_INSTS_VALU") else None,
                    "source": prof.get("source")}
            roof["traffic"] = prof.get("hbm_bytes_per_frame")
            roof["traffic_source"] = prof.get("hbm_source")
        if algo:
            bytes_launch = algo["bytes"]
            gbs = bytes_launch / (kern_ms * 1e-3) / 1e9
            roof["bytes"] = {"algorithmic_bytes_per_launch": bytes_launch, "bytes_per_ray": round(bytes_launch / rays_rank, 1),
                             "algorithmic_GBs": round(gbs, 1),
                             "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "hbm_peak_GBs": HBM_PEAK_GBS,
                             "frac_of_l2_peak": round(gbs / L2_PEAK_GBS, 4), "l2_peak_GBs": L2_PEAK_GBS,
                             "note": "SURVEY s8d formula (52 B per node / instance record, 36 B per triangle the reference would fetch, + shading bytes); the scene is "
                                     "cache-resident, so this figure prices bytes the L1/L2/Infinity Cache serve: it may exceed the HBM peak and is not the bound",
                             "measured_hbm_GBs": round(roof["traffic"] / (kern_ms * 1e-3) / 1e9, 1) if roof.get("traffic") else None,
                             "measured_hbm_frac_of_peak": round(roof["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if roof.get("traffic") else None}
            roof["counts_reference_order"] = algo
            roof["counts_timed_traversal"] = algo_timed
        out["roofline"] = roof
        rr = extras.get("random_rays")
        if rr:
            # north_star's second figure with its own roofline block: N random rays per GPU against the fixed BVH, as absolute, against
            # the VALU issue roof (numerator: SQ_INSTS_VALU of the ray-buffer kernel's launch, a profile constant tied to this tree, this
            # kernel source and these rays by the fetch counts the counting build reports) and, as north_star words it, the SURVEY s8d
            # bytes against the HBM roofline
            blk = {"bound": "valu", "kernel": "rt_persistent_kernel<JOB_TRACE> (+ its EXACT launch)", "rays_per_gpu": rr["rays_per_gpu"], "n_gpus": world,
                   "mrays_s": rr["mrays_s"], "ms_per_launch": rr["ms_per_launch"], "achieved": None, "peak": round(world * SIMDS * clock / 2.0, 1),
                   "unit": "G wave64 VALU instructions/s", "frac": None,
                   "bytes": {"bytes_per_ray": rr["bytes_per_ray"], "algorithmic_GBs": rr["algorithmic_GBs"], "hbm_peak_GBs": HBM_PEAK_GBS * world,
                             "frac_of_hbm_peak": rr["frac_of_hbm_peak"], "frac_of_l2_peak": rr["frac_of_l2_peak"]}}
            pr = prof.get("random_rays") if prof else None
            why_rr = why_not
            if not why_rr and not pr:
                why_rr = "profiles/valu_profile.json holds no counter pass of the ray-buffer kernel"
            near = lambda a, b: bool(a and b) and abs(a - b) <= 1e-3 * max(a, b)      # (see profile_mismatch)
            if not why_rr and (pr.get("n") != rr["rays_per_gpu"] or not near(pr.get("node_fetches"), rr["node_fetches"]) or not near(pr.get("tri_fetches"), rr["tri_fetches"])):
                why_rr = "the profiled ray buffer differs from this run's (rays or fetch counts)"
            if why_rr:
                blk["frac_is_null_because"] = why_rr
            else:
                ach = pr["valu_instr_per_launch"] * world / (rr["ms_per_launch"] * 1e-3) / 1e9
                blk.update({"achieved": round(ach, 1), "frac": round(ach / (world * SIMDS * clock / 2.0), 4), "valu_instr_per_launch": pr["valu_instr_per_launch"],
                            "lane_utilisation": pr.get("lane_utilisation"), "wait_any_of_wave_cycles": pr.get("wait_any_of_wave_cycles"),
                            "valu_source": pr.get("source")})
            out["roofline_random_rays"] = blk
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, vrt, W, H, LIGHT, a.cpu_seconds, random_sample)
        if extras:
            out["extras"] = extras
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
