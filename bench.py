#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ray-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s (primary + shadow) at 1920x1080 on the 1M-triangle "Sponza-class"
BVH.  A step = one frame of the RTU test on that scene: camera ray -> closest hit -> Lambert shade
with one occlusion ray toward the light per hit -> RGB8, i.e. one vxrt_render call through the C ABI
(persistent traversal launch + EXACT launches + shading pass) on this rank's GPU, scene already resident in
HBM.  Frames are issued round robin on --frames-in-flight streams (default 4) so that the draining tail of
one frame's persistent launch overlaps the next frame's; --frames-in-flight 1 gives strictly serial frames.

Multi-GPU (one process per GPU, torch.distributed over RCCL): the reference's own per-pixel
`for s < samples_per_pixel` loop (kernel.cpp:67-80) is the data-parallel axis -- rank r traces
sample r of every pixel (weak scaling: one full frame of rays per GPU and step, no data-path
collective), then ONE gather over xGMI assembles the per-sample frames on rank 0 (north_star:
"RCCL gather only for final image assembly").  `--shard rows` instead splits ONE frame into
tile-aligned row bands (strong scaling).  value = rays traced by all ranks / max-over-ranks time.

Prints one JSON line (driver contract) with `roofline` and `cpu_baseline` objects.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--level", type=int, default=8, help="atrium tessellation level; 8 -> 1,048,576 triangles")
    ap.add_argument("--shard", choices=["samples", "rows"], default="samples")
    ap.add_argument("--no-shadow", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-seconds of host work for cpu_baseline")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="frames kept in flight on as many HIP streams (vxrt_accel_frames_in_flight); 1 = strictly serial frames")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal of the N>1 code path where ranks share one GPU (frames gathered through host memory)")
    ap.add_argument("--random-rays", type=int, default=16777216,
                    help="second leg (SURVEY s8d 'random rays vs fixed BVH'): N incoherent rays per GPU through vxrt_trace, reported under extras; 0 = skip")
    return ap.parse_args()


def cpu_baseline(scene, w, h, budget_cpu_s):
    """The reference's own BVHTraverser (oracle/_ref, kind 'reference') -- or, if that library is
    absent or exceeds its watchdog, the C restatement (kind 'port') -- timed on this box's host cores
    over a bounded sample of the same workload: the camera rays of the frame (closest hit only; the
    reference has no shadow rays), repeated until about `budget_cpu_s` CPU-seconds are spent.
    Checker code is used here as the thing timed for the reported baseline only, never for `value`."""
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

    def run(fn, rays, repeats):
        chunks = np.array_split(np.arange(len(rays)), cores * 8)
        t0 = time.perf_counter()
        done = 0
        with cf.ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL during the foreign call
            futs = [ex.submit(fn, img, rays[c]) for _ in range(repeats) for c in chunks]
            for f in futs:
                try:
                    out, _ = f.result(timeout=max(5.0, 8 * budget_cpu_s / cores - (time.perf_counter() - t0)))
                    done += len(out)
                except cf.TimeoutError:
                    return None
        return done, time.perf_counter() - t0

    def sized(fn):
        t0 = time.perf_counter()
        fn(img, rays[:: max(1, len(rays) // 4000)])          # probe spread over the frame
        per_ray = (time.perf_counter() - t0) / len(rays[:: max(1, len(rays) // 4000)])
        want = budget_cpu_s / per_ray                        # rays for the CPU-second budget
        if want >= len(rays):
            return rays, max(1, int(round(want / len(rays))))
        return rays[:: max(1, int(len(rays) / want))], 1

    kind, res = "port", None
    if po.have_ref():
        r, rep = sized(po.trace_ref)
        res = run(po.trace_ref, r, rep)
        kind = "reference" if res else "port"
    if res is None:
        r, rep = sized(po.trace_faithful)
        res = run(po.trace_faithful, r, rep)
    done, dt = res
    return {"value": round(done / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": "%d primary camera rays of the %dx%d frame (closest hit, no shadow rays; %d x %d rays), %s, %d host threads, %.1f s wall"
                      % (done, w, h, rep, len(r), "reference sim/simx/rt_traversal.cpp via oracle/_ref" if kind == "reference" else "oracle/rt_oracle.c restatement", cores, dt)}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback for the hot path")
    if a.dist_backend == "gloo":
        local %= max(1, torch.cuda.device_count())   # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if a.dist_backend == "nccl" else "cpu"   # where collectives' tensors live

    vrt = importlib.import_module("vortex-raytracing_amd")
    rtapi, sharding = vrt.rtapi, vrt.sharding

    W, H = a.width, a.height
    scene = vrt.scene.procedural("atrium", a.level, 0, 3)
    ds = vrt.tracer.DeviceScene(scene, dev)
    shadow = 0 if a.no_shadow else 1
    params = rtapi.default_shade_params()
    params.light_pos[:] = (300.0, 480.0, 60.0)   # inside the hall, so shadow rays are real work

    if a.shard == "rows" and world > 1:
        y0, y1 = sharding.row_bands(H, world)[rank]
    else:
        y0, y1 = 0, H
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    nfl = max(1, min(8, a.frames_in_flight))
    # frame i goes to stream i % nfl and framebuffer i % nfl; the accel keeps nfl frame contexts
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(nfl - 1)]
    # (with several ranks a framebuffer stays busy until its gather has run: twice as many, so that rendering never waits for the link)
    frames = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(max(2, nfl) * (2 if world > 1 else 1))]
    counters = torch.zeros(8, dtype=torch.int64, device=dev)

    def launch(buf, count_ptr=None, st=None):
        rtapi.render(ds.accel, W, H, y0, y1, params, buf.data_ptr(), shadow, None, None, count_ptr, (st or stream).cuda_stream)

    # rays per step on this rank (primary + shadow), counted once by the kernel itself
    launch(frames[0], counters.data_ptr())
    torch.cuda.synchronize()
    assert rtapi.status(sptr) == 0, "kernel status (traversal stack overflow)"
    rays_rank = int(counters[0].item())
    algo = None
    if hasattr(rtapi, "render_stats"):
        algo = rtapi.render_stats(ds.accel, W, H, y0, y1, params, frames[0].data_ptr(), shadow, sptr)

    rtapi.accel_frames_in_flight(ds.accel, nfl)
    gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    gdone = [None] * len(frames)   # per framebuffer: event of the last gather that read it

    def step(i, ev=None):
        b = i % len(frames)
        buf = frames[b]
        st = streams[i % nfl]
        if gdone[b] is not None:
            st.wait_event(gdone[b])            # do not overwrite a frame that is still being gathered
        if ev is not None:
            ev[0].record(st)
        launch(buf, st=st)
        if ev is not None:
            ev[1].record(st)
        if world > 1:
            # image assembly overlaps the next step's traversal: the gather runs on its own stream
            gather_stream.wait_stream(st)
            with torch.cuda.stream(gather_stream):
                src = buf if cdev != "cpu" else buf.cpu()
                if a.shard == "rows":
                    sharding.gather_frame(src[y0:y1], H, W, rank, world)
                else:
                    out = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
                    dist.gather(src, out, dst=0)
                gdone[b] = torch.cuda.Event()
                gdone[b].record(gather_stream)

    # isolated duration of one step's launches (nothing else on the GPU): HIP events on the launch stream
    iso = []
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    for i in range(min(10, max(3, a.steps))):
        e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e[0].record(stream)
        launch(frames[0])
        e[1].record(stream)
        torch.cuda.synchronize()
        iso.append(e[0].elapsed_time(e[1]))
    iso_ms = sum(iso) / len(iso)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_rank], dtype=torch.int64, device=cdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_all = int(r.item())
    else:
        rays_all = rays_rank
    assert rtapi.status(sptr) == 0
    # HIP events on the launch streams.  With frames in flight the launches of consecutive steps overlap, so
    # the per-launch duration that prices the roofline is the span of the timed region's events divided by
    # the launches in it (their union, not their sum); the overlapped and the isolated per-launch
    # durations are reported next to it.
    ovl_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / a.steps
    span_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs)
    kern_ms = span_ms / a.steps

    extras = {}
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
        hits = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
        rstats = rtapi.trace_stats(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr) if rank == 0 else None
        reps = 5
        for _ in range(2):
            rtapi.trace(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            rtapi.trace(ds.accel, rays.data_ptr(), n, hits.data_ptr(), rtapi.MODE_CLOSEST, None, sptr)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        rt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([rt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rt = float(t.item())
        assert rtapi.status(sptr) == 0
        mr = n * world * reps / rt / 1e6
        extras["random_rays_mrays_s"] = round(mr, 1)
        extras["random_rays_n"] = n * world
        if rstats:
            gbs = mr * 1e6 * rstats["bytes_per_ray"] / 1e9
            extras["random_rays"] = {"rays_per_gpu": n, "mrays_s": round(mr, 1), "ms_per_launch": round(rt / reps * 1e3, 3),
                                     "bytes_per_ray": round(rstats["bytes_per_ray"], 1), "achieved_GBs": round(gbs, 1),
                                     "frac_of_hbm_peak": round(gbs / (HBM_PEAK_GBS * world), 4),
                                     "node_fetches_per_ray": round(rstats["node_fetches"] / n, 2),
                                     "tri_fetches_per_ray": round(rstats["tri_fetches"] / n, 2)}
        del rays, hits

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+shadow) at %dx%d, 1M-tri BVH" % (W, H),
            "value": round(rays_all * a.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if (a.shard == "rows" and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (procedural 'Sponza-class' atrium, seed 3; no Sponza/bunny asset exists offline)",
            "config": {"workload": "configs[2]: Sponza-class %d tris, %dx%d, primary + 1 shadow ray per hit%s" % (scene.n_tris, W, H, "" if shadow else " (shadow disabled)"),
                       "rays_per_step_per_gpu": rays_rank, "frames_in_flight": nfl,
                       "parallelism": ("spp-sharded x%d: one sample (full frame) per GPU, RCCL gather of frames to rank 0" % world) if a.shard == "samples"
                                      else ("row bands x%d of one frame, RCCL gather to rank 0" % world),
                       "bvh_nodes": scene.n_bvh_nodes, "bvh_depth": scene.info.get("max_depth")},
        }
        bytes_launch = None
        if algo is not None:
            bytes_launch = algo["bytes"]
        # one step = the launches of vxrt_render: persistent traversal kernel (dominant, > 93 % of the step), the
        # EXACT launches for the rays with NaN-capable slabs, and the shading pass; priced together
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "rt_persistent_kernel<JOB_RENDER%s> (+ EXACT launches + rt_shade_kernel)" % ("_SHADOW" if shadow else ""),
                "kernel_ms": round(kern_ms, 4), "kernel_ms_overlapped": round(ovl_ms, 4), "kernel_ms_isolated": round(iso_ms, 4),
                "frames_in_flight": nfl,
                "note": "algorithmic bytes are SURVEY s8d's per-ray formula (52 B per node, 36 B per triangle the reference would fetch); the scene is "
                        "cache-resident (traffic = measured HBM bytes per step), so achieved can exceed the HBM peak; the kernel's own limit is VALU issue "
                        "(88 % for an isolated launch, 69 % active lanes: profiles/r01_k_pmc.txt, DESIGN.md s4)"}
        if bytes_launch:
            ach = bytes_launch / (kern_ms * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                         "frac_isolated": round(bytes_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": bytes_launch,
                         "bytes_per_ray": round(bytes_launch / rays_rank, 1), "counts": algo})
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                roof["traffic"] = json.load(open(tf)).get("bytes_per_launch")
            except Exception:
                pass
        out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, a.cpu_seconds)
        if extras:
            out["extras"] = extras
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
