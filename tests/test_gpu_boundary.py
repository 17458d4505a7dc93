"""GPU: the drop-in boundary.  The host-side mirror of the reference Tracer drives the HIP backend
through vx_dev_open / vx_mem_alloc / vx_copy_to_dev / vx_dcr_write / vx_start / vx_ready_wait only
(the calls of tests/regression/raytracing/tracer.cpp:114-166,217-259,272-281)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rtu_test_frame_through_vx_api(vrt, po, gpu_device):
    sc = vrt.scene.procedural("blob", 4, 0, 1)
    w, h = 192, 136
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    tr.setup()
    px = tr.run()
    rpx, rhits, _ = po.render(sc, w, h)
    assert np.array_equal(px, rpx)
    # perf counters the reference dumps at close (stub/perf.cpp:195-227): rays and cycles of the run
    assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) == w * h
    assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MCYCLE, 0) > 0
    assert tr.dev.caps(vrt.runtime.VX_CAPS_NUM_THREADS) == 64
    assert tr.dev.caps(vrt.runtime.VX_CAPS_NUM_CORES) >= 1
    # second run on the same device, different light: start waits for the previous run (vortex.cpp:331-333)
    tr.setup(light_pos=(100.0, 300.0, 50.0))
    px2 = tr.run()
    rpx2, _, _ = po.render(sc, w, h, po.shade_params(light_pos=(100.0, 300.0, 50.0)))
    assert np.array_equal(px2, rpx2)
    assert not np.array_equal(px, px2)
    tr.close()


def test_mcycle_is_the_runs_time_on_the_device(vrt, gpu_device):
    """vx_mpm_query(MCYCLE) (stub/perf.cpp:195-227 divides by it) = the run's duration x the shader clock, and the duration is taken on the
    device (a stamp kernel vx_start launches on a stream of its own -> the run's last kernel, 100 MHz clock): a host that sleeps between
    vx_start and vx_ready_wait does not lengthen it, while the host-side clock of the same run (stat 6) does see the sleep.  Both kernels of
    the boundary: the RTU frame and the software twin (reference-quirks runs and runs split over VORTEX_HIP_DEVICES: their own tests)."""
    import time
    w, h = 320, 200
    for twin in (False, True):
        if twin:
            tr = vrt.tracer.RaycastTracer(w, h)
            tr.init(vrt.scene.rc_procedural("cornell"))
            tr.setup(vrt.scene.rc_camera_like_rtu(w, h), (0.0, 10.0, -10.0, 1.0, 1.0, 1.0, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25))
        else:
            tr = vrt.tracer.Tracer(w, h)
            tr.init(vrt.scene.procedural("blob", 4, 0, 1))
            tr.setup()
        tr.run()                                         # (first run: layout build, code load)
        d = tr.dev
        n_dev0 = d.hip_stat(4)
        d.start(tr.krnl, tr.args)
        d.ready_wait(vrt.runtime.VX_MAX_TIMEOUT)
        quick = d.mpm_query(vrt.runtime.VX_CSR_MCYCLE, 0)
        d.start(tr.krnl, tr.args)
        time.sleep(0.25)                                 # the host is busy elsewhere while the device runs and finishes
        d.ready_wait(vrt.runtime.VX_MAX_TIMEOUT)
        slept = d.mpm_query(vrt.runtime.VX_CSR_MCYCLE, 0)
        host_us = d.hip_stat(6)
        assert d.hip_stat(4) == n_dev0 + 2 and d.hip_stat(5) == 0
        assert host_us >= 250000                          # the host's clock saw the sleep ...
        assert quick > 0 and slept > 0
        assert slept < 5 * 10**7                          # ... the device's did not: 250 ms at >= 1 GHz would be > 2.5e8 cycles
        assert slept < 50 * quick                         # and the two runs of one frame took comparable time on the device
        tr.close()


def test_row_window_dcrs_shard_a_frame(vrt, po, gpu_device):
    """Backend extension DCRs 0x7F0/0x7F1: each 'rank' renders its row band; bands tile the frame."""
    sc = vrt.scene.procedural("cornell")
    w, h = 80, 64
    rpx, _, _ = po.render(sc, w, h)
    frame = np.zeros((h, w), np.uint32)
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    for (y0, y1) in vrt.sharding.row_bands(h, 3):
        tr.setup(row_window=(y0, y1))
        band = tr.run()
        frame[y0:y1] = band[y0:y1]
    tr.close()
    assert np.array_equal(frame, rpx)


def test_error_behaviour_of_the_boundary(vrt, gpu_device):
    rt = vrt.runtime
    L = rt.lib()
    d = rt.Device()
    b = C.c_void_p()
    assert L.vx_mem_alloc(d.handle, 0, rt.VX_MEM_READ, C.byref(b)) != 0          # callbacks.inc:61-65
    assert L.vx_mem_alloc(None, 64, rt.VX_MEM_READ, C.byref(b)) != 0
    assert L.vx_mem_free(None) == 0                                                # callbacks.inc:100-102
    buf = d.mem_alloc(100)
    data = (C.c_uint8 * 128)()
    assert L.vx_copy_to_dev(buf.handle, data, 0, 101) != 0                         # bounds (callbacks.inc:153)
    assert L.vx_copy_to_dev(buf.handle, data, 64, 64) != 0
    assert L.vx_copy_to_dev(buf.handle, data, 36, 64) == 0
    assert L.vx_copy_from_dev(data, buf.handle, 90, 11) != 0
    assert L.vx_copy_to_dev(buf.handle, None, 0, 4) != 0
    v = C.c_uint32()
    assert L.vx_dcr_read(d.handle, 0x123, C.byref(v)) != 0                         # unwritten DCR (common.h:60-66)
    d.dcr_write(0x123, 77)
    assert d.dcr_read(0x123) == 77
    assert L.vx_ready_wait(d.handle, 10) == 0                                      # nothing running
    # addresses follow the simx allocator convention: 64-byte blocks from USER_BASE_ADDR
    b2 = d.mem_alloc(1)
    assert buf.address >= 0x10000 and buf.address % 64 == 0 and b2.address % 64 == 0 and b2.address != buf.address
    # a RISC-V image cannot be started: start must fail loudly, not fall back to anything
    blob = struct.pack("<QQ", 0x80000000, 0x80001000) + b"\x13\x00\x00\x00" * 16
    k = C.c_void_p()
    assert L.vx_upload_kernel_bytes(d.handle, blob, len(blob), C.byref(k)) == 0
    args = d.upload_bytes(bytes(216))
    assert L.vx_start(d.handle, k, args.handle) != 0
    # reserved range cannot be reserved twice
    k2 = C.c_void_p()
    assert L.vx_upload_kernel_bytes(d.handle, blob, len(blob), C.byref(k2)) != 0
    for h in (buf, b2, args):
        h.free()
    assert L.vx_mem_free(k) == 0
    d.close()


def test_missing_dcrs_or_wrong_sbt_fail(vrt, gpu_device):
    sc = vrt.scene.procedural("cornell")
    tr = vrt.tracer.Tracer(32, 32)
    tr.init(sc)
    tr.setup()
    # corrupt the shader binding table: closest-hit entry points at the miss selector
    tr.bufs["sbt"].write(struct.pack("<4Q", tr.miss.address, tr.miss.address, 0, tr.anyhit.address))
    with pytest.raises(vrt.runtime.VxError):
        tr.run()
    tr.setup(background=(0.1, 0.2, 0.3))
    assert tr.run().shape == (32, 32)
    tr.close()


def test_host_program_renders_and_matches(vrt, po, gpu_device, tmp_path):
    """The C++ host program (csrc/rt_host.cpp), same CLI as the reference's main.cpp, end to end."""
    import subprocess
    exe = os.path.join(vrt.LIB_DIR, "rt_host")
    if not os.path.exists(exe):
        pytest.skip("rt_host not built")
    out = tmp_path / "out.ppm"
    env = dict(os.environ, LD_LIBRARY_PATH=vrt.LIB_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""), VORTEX_DRIVER="hip")
    r = subprocess.run([exe, "-m", "proc:cornell", "-w", "48", "-h", "40", "-o", str(out), "-k", os.path.join(vrt.VXBIN_DIR, "kernel.vxbin")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = np.array(out.read_text().split()[4:], dtype=np.int64).reshape(40, 48, 3)
    rpx, _, _ = po.render(vrt.scene.procedural("cornell"), 48, 40)
    want = np.stack([(rpx >> 16) & 255, (rpx >> 8) & 255, rpx & 255], -1)[::-1]
    assert np.array_equal(vals, want)


def test_samples_per_pixel_traces_the_frame_that_many_times(vrt, po, gpu_device):
    """kernel_arg_t::samples_per_pixel as the reference's kernel treats it (kernel.cpp:67-80): the same camera ray traced spp times into the
    same payload -- the pixels of one sample, spp times the rays (MINSTRET).  spp = 0 is refused (pixels undefined in the reference)."""
    sc = vrt.scene.procedural("blob", 3, 0, 2)
    w, h = 72, 48
    px = {}
    rays = {}
    for spp in (1, 3):
        tr = vrt.tracer.Tracer(w, h, samples_per_pixel=spp)
        tr.init(sc)
        tr.setup()
        px[spp] = tr.run()
        rays[spp] = tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0)
        tr.close()
    want, _, _ = po.render(sc, w, h)
    assert np.array_equal(px[1], want) and np.array_equal(px[3], want)
    assert rays[1] == w * h and rays[3] == 3 * w * h


def test_small_scene_buffers_reach_the_device_before_they_are_read(vrt, po, gpu_device):
    """Uploads of up to 4 KB are kept in the host shadow and sent to the device when something there reads them (the kernel arguments of
    every frame never are).  A 12-triangle scene's own buffers ARE that small: re-uploaded between two runs, the second run must see the
    new bytes -- and vx_copy_from_dev of such a buffer returns what was uploaded."""
    sc = vrt.scene.procedural("cornell")
    w, h = 40, 32
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    tr.setup()
    first = tr.run()
    want, _, _ = po.render(sc, w, h)
    assert np.array_equal(first, want)
    assert sc["blas"].size <= 4096 and sc["tlas"].size <= 4096
    # move the instance 3 units along z by rewriting its record (transform + inverse), as a host animating the scene would
    blas = sc["blas"].copy().view(np.float32)
    blas[17 + 11] += 3.0       # transform[2][3]
    blas[1 + 11] -= 3.0        # invTransform[2][3]
    tr.bufs["blas"].write(blas.view(np.uint8))
    assert bytes(tr.bufs["blas"].read()) == blas.view(np.uint8).tobytes()
    moved = tr.run()
    sc2 = vrt.scene.Scene(dict(sc.buffers, blas=blas.view(np.uint8)))
    want2, _, _ = po.render(sc2, w, h)
    assert np.array_equal(moved, want2) and not np.array_equal(moved, first)
    tr.close()


def test_a_small_framebuffer_cleared_by_the_host_keeps_the_rendered_pixels(vrt, po, gpu_device):
    """A framebuffer of <= 4 KB (16x16 pixels) that the host writes before vx_start -- clearing it, as a frame loop does -- is a lazy
    upload like any other small buffer.  It must reach the device BEFORE the run: flushed by the vx_copy_from_dev after the run, it
    would put the host's zeros over the pixels the kernels just wrote.  Both kernels (RTU frame, twin); a row window keeps the
    host's bytes outside it."""
    sc = vrt.scene.procedural("cornell")
    w, h = 16, 16
    want, _, _ = po.render(sc, w, h)
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    tr.setup()
    tr.bufs["out"].write(np.zeros(w * h * 4, np.uint8))
    assert np.array_equal(tr.run(), want)
    # second frame of the loop: cleared to another pattern, rendered again
    tr.bufs["out"].write(np.full(w * h * 4, 0xAB, np.uint8))
    assert np.array_equal(tr.run(), want)
    # a row window: rows outside it keep what the host uploaded, rows inside are the frame's
    tr.bufs["out"].write(np.full(w * h * 4, 0xCD, np.uint8))
    tr.setup(row_window=(8, 16))
    got = tr.run()
    assert np.array_equal(got[8:], want[8:]) and np.all(got[:8] == 0xCDCDCDCD)
    tr.close()
    rsc = vrt.scene.rc_procedural("cornell")
    rt = vrt.tracer.RaycastTracer(w, h)
    rt.init(rsc)
    rt.setup(vrt.scene.rc_camera_like_rtu(w, h), (0.0, 10.0, -10.0, 1.0, 1.0, 1.0, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25))
    first = rt.run().copy()
    rt.bufs["out"].write(np.zeros(w * h * 4, np.uint8))
    assert np.array_equal(rt.run(), first) and np.any(first != 0)
    rt.close()


def test_mirror_bounce_through_vx_api(vrt, po, gpu_device):
    """kernel_arg_t::max_depth + blas_node_t::reflectivity reach the kernels through vx_copy_to_dev /
    vx_start exactly as the reference host passes them (tracer.cpp:217-259, main.cpp -d)."""
    from scenes import mirror_hall
    b = mirror_hall(vrt)
    w, h = 128, 80
    tr = vrt.tracer.Tracer(w, h, max_depth=3)
    tr.init(b)
    tr.setup(light_pos=(150.0, 220.0, -60.0))
    px = tr.run()
    rpx, _, _, rn = po.render_ex(b, w, h, po.shade_params(light_pos=(150.0, 220.0, -60.0), max_depth=3))
    assert np.array_equal(px, rpx)
    assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) == rn > w * h
    flat, _, _ = po.render(b, w, h, po.shade_params(light_pos=(150.0, 220.0, -60.0)))
    assert not np.array_equal(px, flat)
    tr.close()


def test_mirror_bounce_through_vx_api_matches_the_reference_twin(vrt, golden, gpu_device):
    """The same pin through the drop-in boundary: the reference-built RTU buffers of tests/golden/mirror_trio.npz (three instances,
    reflectivity 0 / 0.5 / 0.3 in their records) uploaded with vx_copy_to_dev, kernel_arg_t::max_depth = 3, vx_start: the pixels
    equal what the reference's software twin (raycast/render.h) returns for these camera rays."""
    g = golden("mirror_trio")
    w, h = int(g["width"]), int(g["height"])
    L = g["light12"]
    g = dict(g, triIdx=np.arange(g["tri"].size // 36, dtype=np.uint32).view(np.uint8))   # (kernel_arg_t names one; the RTU path never reads it)
    tr = vrt.tracer.Tracer(w, h, max_depth=3)
    tr.init(g)
    tr.setup(light_pos=tuple(L[0:3]), light_color=tuple(L[3:6]), ambient=tuple(L[6:9]), background=tuple(L[9:12]))
    px = tr.run()
    assert np.array_equal(px, g["rgb8_d3"]), "%d pixels differ" % int((px != g["rgb8_d3"]).sum())
    assert not np.array_equal(px, g["rgb8_d1"])
    tr.close()


def test_runs_do_not_rebuild_or_reallocate(vrt, gpu_device):
    """The reference host re-uploads kernel_arg_t for every run (tracer.cpp:272-281: vx_upload_bytes -> a new buffer) and
    our mirror frees the previous one: neither may cost an acceleration-layout build or a hipMalloc per run."""
    sc = vrt.scene.procedural("blob", 3, 0, 1)
    tr = vrt.tracer.Tracer(96, 64)
    tr.init(sc)
    tr.setup()
    first = tr.run()
    builds, mallocs = tr.dev.hip_stat(0), tr.dev.hip_stat(1)
    assert builds == 1
    for _ in range(5):
        assert np.array_equal(tr.run(), first)
    assert tr.dev.hip_stat(0) == 1 and tr.dev.hip_stat(1) == mallocs
    tr.setup(light_pos=(100.0, 300.0, 50.0))     # re-uploads the scene buffers: one rebuild, no new device allocations
    tr.run()
    assert tr.dev.hip_stat(0) == 2 and tr.dev.hip_stat(1) == mallocs
    tr.close()


def test_kernel_arguments_through_the_staging_ring_stay_in_order(vrt, po, gpu_device):
    """vx_upload_bytes of the 216-byte kernel_arg_t goes through a pinned ring of 64 slots on the run's stream (no synchronous copy):
    200 runs alternating between two argument blocks -- past several wraps of the ring -- each render the frame of THEIR block."""
    sc = vrt.scene.procedural("blob", 3, 0, 1)
    w, h = 96, 64
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    lights = ((300.0, 480.0, 60.0), (100.0, 300.0, 50.0))
    want, args = [], []
    for lp in lights:
        tr.setup(light_pos=lp)
        args.append(tr.kernel_arg)
        want.append(tr.run().copy())
        rpx, _, _ = po.render(sc, w, h, po.shade_params(light_pos=lp))
        assert np.array_equal(want[-1], rpx)
    assert not np.array_equal(want[0], want[1])
    mallocs = tr.dev.hip_stat(1)
    for i in range(200):
        k = (i * 7 + i // 3) & 1
        tr.kernel_arg = args[k]
        assert np.array_equal(tr.run(), want[k]), i
    assert tr.dev.hip_stat(1) == mallocs
    tr.close()


def test_reference_quirks_dcr_renders_what_the_rtu_would_on_this_address_space(vrt, po, gpu_device):
    """DCR 0x7F4 = 1 through the vx_* boundary: the frame is traced by the literal restatement of the reference RTU (stale base_ptr
    of rt_traversal.cpp:91-92 included) on a flat image of the device's address space, with the RTX DCR values as base pointers.
    Expected pixels: the oracle's faithful restatement on the SAME address space (every buffer at the address vx_mem_address
    reported), shaded by the oracle.  The scene has 12 instances -- a TLAS deeper than one level -- in the camera's view, so the
    quirk is live: the frame differs from the canonical one (DCR 0x7F4 = 0, the default), which equals the canonical oracle."""
    rng = np.random.default_rng(5)
    base = vrt.scene.procedural("blob", 2, 0, 1)["tri"].view(np.float32).reshape(-1, 9)
    base = (base - base.reshape(-1, 3).mean(0).tolist() * 3) * np.float32(0.25)
    xf = []
    for i in range(12):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = (260.0 + 45.0 * (i % 3) + 130.0 * (i // 6), 100.0 + 40.0 * ((i // 3) % 2), 60.0 * (i % 3 - 1))
        xf.append(m)
    sc = vrt.scene.from_triangles([base] * 12, xf)
    assert sc.n_tlas_nodes > 5                      # internal TLAS nodes below the root
    w, h = 160, 96
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    tr.setup()
    canon = tr.run()
    want_canon, _, _ = po.render(sc, w, h)
    assert np.array_equal(canon, want_canon)
    tr.dev.dcr_write(0x7F4, 1)
    n_dev0 = tr.dev.hip_stat(4)
    quirk = tr.run()
    assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MCYCLE, 0) > 0 and tr.dev.hip_stat(4) == n_dev0 + 1 and tr.dev.hip_stat(5) == 0   # (the device's clock in this mode too)
    tr.dev.dcr_write(0x7F4, 0)
    again = tr.run()
    assert np.array_equal(again, canon)             # the mode is a switch, nothing sticks
    # the address space as the backend laid it out
    addr = {k: b.address for k, b in tr.bufs.items()}
    top = max(addr[k] + int(sc[k].size) for k in ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "mat", "tex"))
    assert top < 2 ** 32
    mem = np.zeros(top + 64, np.uint8)
    for k in ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "mat", "tex"):
        b = np.ascontiguousarray(sc[k], np.uint8).reshape(-1)
        mem[addr[k]: addr[k] + b.size] = b

    class Img:   # what pyoracle.Image is, on the device's layout
        pass
    img = Img()
    img.mem, img.off = mem, {k: addr[k] for k in ("tlas", "blas", "bvh", "tri")}
    img.__class__ = type("Image", (po.Image,), {})
    rays = po.camera_rays(w, h)
    hits, st = po.trace_faithful(img, rays)
    _, want_quirk = po.shade(sc, rays, hits)
    assert np.array_equal(quirk, want_quirk.reshape(h, w))
    tr.close()
    print("quirk mode: %d rays took the stale base_ptr, %d of %d pixels differ from the canonical frame" % (st["stale_base"], int((quirk != canon).sum()), w * h))
    assert st["stale_base"] > 0 and (quirk != canon).any()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,shape", [("0,0", (200, 120)), ("0,0,0", (136, 61)), ("0,0,0,0,0", (64, 40))])
def test_vortex_hip_devices_splits_a_frame_behind_one_vx_device(vrt, po, gpu_device, monkeypatch, devices, shape):
    """VORTEX_HIP_DEVICES=a,b,...: the unmodified reference host opens ONE device (tracer.cpp:78); the backend keeps the address space on the
    first listed GPU and has every listed GPU trace tile rows k, k+n, ... of each whole frame from its own copy of the scene.  On the one-GPU
    box the list repeats device 0 (every share has its own copy, layout, stream and framebuffer, as on separate GPUs; only the copies stay
    on one card).  The frame, the ray count (MINSTRET) and a re-uploaded scene buffer must behave as with one device; a run the host
    restricts with the row-window DCRs stays on the first device.  (136x61: the frame's last tile row is 5 rows high.)"""
    w, h = shape
    sc = vrt.scene.procedural("blob", 3, 0, 2)
    want = po.render_ex(sc, w, h, shadow=1)[0]
    n = len(devices.split(","))
    monkeypatch.setenv("VORTEX_HIP_DEVICES", devices)
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    monkeypatch.delenv("VORTEX_HIP_DEVICES")
    tr.setup(shadow=True)
    assert tr.dev.hip_stat(3) == n
    px = tr.run()
    assert np.array_equal(px, want)
    rays = tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0)
    one = vrt.tracer.Tracer(w, h)
    one.init(sc)
    one.setup(shadow=True)
    assert np.array_equal(one.run(), want)
    assert rays == one.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) and one.dev.hip_stat(3) == 1
    one.close()
    assert tr.dev.hip_stat(2) == 1 and tr.dev.hip_stat(0) == n     # one layout per share, built once
    for _ in range(3):
        assert np.array_equal(tr.run(), want)
    assert tr.dev.hip_stat(2) == 4 and tr.dev.hip_stat(0) == n
    # a re-uploaded scene buffer reaches every share: move the instance and compare with the oracle on the moved scene
    blas = sc["blas"].copy().view(np.float32)
    blas[17 + 3] += 6.0        # transform[0][3]
    blas[1 + 3] -= 6.0         # invTransform[0][3]
    tr.bufs["blas"].write(blas.view(np.uint8))
    sc2 = vrt.scene.Scene(dict(sc.buffers, blas=blas.view(np.uint8)))
    want2 = po.render_ex(sc2, w, h, shadow=1)[0]
    assert not np.array_equal(want2, want)
    assert np.array_equal(tr.run(), want2)
    # a row window set by the host: that run is the first device's alone, and renders only the window (setup() uploads the original scene again)
    before = tr.dev.hip_stat(2)
    tr.setup(shadow=True, row_window=(8, 24))
    part = tr.run()
    assert tr.dev.hip_stat(2) == before
    assert np.array_equal(part[8:24], want[8:24])
    tr.close()


@pytest.mark.gpu
def test_vortex_hip_devices_gathered_through_rccl(vrt, po, gpu_device, monkeypatch):
    """VORTEX_HIP_GATHER=rccl (north_star: "host code stays in C ... RCCL gather over xGMI only for final image assembly"): the shares of a frame
    split over VORTEX_HIP_DEVICES reach the first device's output buffer through ONE group of ncclSend / ncclRecv pairs issued by the C++
    backend itself (librccl.so.1 bound with dlopen, one communicator per distinct GPU: ncclCommInitAll) instead of peer copies.  RCCL refuses
    the same GPU twice in one communicator, so on the one-GPU box the communicator has ONE rank and the two extra shares -- packed on their own
    streams, from their own framebuffers -- travel as RCCL's self send / receive pairs: the same calls, group, packing and placing as between
    GPUs, and all this box can exercise.  Same frame, same ray count as one device, run after run; ragged last tile row (61 = 7 x 8 + 5)."""
    w, h = 136, 61
    sc = vrt.scene.procedural("blob", 3, 0, 2)
    want = po.render_ex(sc, w, h, shadow=1)[0]
    monkeypatch.setenv("VORTEX_HIP_DEVICES", "0,0,0")
    monkeypatch.setenv("VORTEX_HIP_GATHER", "rccl")
    tr = vrt.tracer.Tracer(w, h)
    tr.init(sc)
    monkeypatch.delenv("VORTEX_HIP_DEVICES")
    monkeypatch.delenv("VORTEX_HIP_GATHER")
    tr.setup(shadow=True)
    assert tr.dev.hip_stat(3) == 3
    for i in range(4):
        tr.bufs["out"].write(np.full(w * h * 4, 0x5A, np.uint8))
        assert np.array_equal(tr.run(), want)
        assert tr.dev.hip_stat(7) == i + 1 and tr.dev.hip_stat(2) == i + 1
    rays = tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0)
    one = vrt.tracer.Tracer(w, h)
    one.init(sc)
    one.setup(shadow=True)
    assert np.array_equal(one.run(), want) and rays == one.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) and one.dev.hip_stat(7) == 0
    one.close()
    # a frame of another size on the same device: the landing buffers grow
    tr.close()
    monkeypatch.setenv("VORTEX_HIP_GATHER", "tcp")
    monkeypatch.setenv("VORTEX_HIP_DEVICES", "0,0")
    with pytest.raises(vrt.runtime.VxError):
        vrt.runtime.Device()
    monkeypatch.delenv("VORTEX_HIP_DEVICES")
    monkeypatch.delenv("VORTEX_HIP_GATHER")


def test_vortex_hip_devices_rejects_a_malformed_list(vrt, gpu_device, monkeypatch):
    for bad in ("0;1", "x", "0,99"):
        monkeypatch.setenv("VORTEX_HIP_DEVICES", bad)
        with pytest.raises(vrt.runtime.VxError):
            vrt.runtime.Device()
    monkeypatch.delenv("VORTEX_HIP_DEVICES")


@pytest.mark.gpu
def test_host_program_on_two_shares(vrt, po, gpu_device, tmp_path):
    """The C++ host (same call sequence as the reference's main.cpp) with VORTEX_HIP_DEVICES set: same picture."""
    import subprocess
    exe = os.path.join(vrt.LIB_DIR, "rt_host")
    if not os.path.exists(exe):
        pytest.skip("rt_host not built")
    out = tmp_path / "out.ppm"
    env = dict(os.environ, LD_LIBRARY_PATH=vrt.LIB_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""), VORTEX_DRIVER="hip", VORTEX_HIP_DEVICES="0,0")
    r = subprocess.run([exe, "-m", "proc:cornell", "-w", "48", "-h", "40", "-o", str(out), "-k", os.path.join(vrt.VXBIN_DIR, "kernel.vxbin")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = np.array(out.read_text().split()[4:], dtype=np.int64).reshape(40, 48, 3)
    rpx, _, _ = po.render(vrt.scene.procedural("cornell"), 48, 40)
    want = np.stack([(rpx >> 16) & 255, (rpx >> 8) & 255, rpx & 255], -1)[::-1]
    assert np.array_equal(vals, want)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "4"))))     # (VXRT_FUZZ_SEEDS=n: a soak run over n seeds)
def test_random_frames_through_the_vx_boundary(vrt, po, gpu_device, seed):
    """The drop-in call sequence on random inputs: frame sizes from 1x1 to 300x200 (ragged last tile rows and columns, frames smaller
    than a tile, framebuffers at and below the 4 KB lazy-upload size), shadow rays on or off, 1-3 samples per pixel, random light, an
    optional row window, an optional cleared framebuffer before the run, two runs on the same device -- every pixel against the oracle,
    MINSTRET against the oracle's ray count times the samples, rows outside a window untouched."""
    rng = np.random.default_rng(52000 + seed)
    sc = vrt.scene.procedural(*[("blob", 3, 0, 2), ("cornell", 0, 0, 1), ("atrium", 3, 0, 3)][int(rng.integers(0, 3))])
    w = int(rng.choice([1, 7, 8, 9, 31, 32, 33, 64, 100, 136, 300]))
    h = int(rng.choice([1, 5, 8, 9, 17, 32, 61, 64, 120, 200]))
    spp = int(rng.integers(1, 4))
    shadow = bool(rng.integers(0, 2))
    tr = vrt.tracer.Tracer(w, h, samples_per_pixel=spp)
    tr.init(sc)
    for run in range(2):
        light = (float(rng.uniform(-100, 500)), float(rng.uniform(50, 500)), float(rng.uniform(-200, 200)))
        window = None
        if h >= 16 and rng.integers(0, 3) == 0:
            y0 = int(rng.integers(0, h // 8)) * 8
            window = (y0, int(rng.integers(y0 + 1, h + 1)))
        fill = int(rng.integers(0, 256))
        tr.bufs["out"].write(np.full(w * h * 4, fill, np.uint8))
        tr.setup(light_pos=light, row_window=window, shadow=shadow)
        got = tr.run()
        y0, y1 = window if window else (0, h)
        want, _, _, nrays = po.render_ex(sc, w, h, po.shade_params(light_pos=light), shadow=int(shadow), y0=y0, y1=y1)
        assert np.array_equal(got[y0:y1], want[y0:y1]), (seed, run, w, h, spp, shadow, window)
        keep = np.uint32(fill * 0x01010101)
        assert (got[:y0] == keep).all() and (got[y1:] == keep).all(), (seed, run, "rows outside the window")
        assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) == spp * nrays, (seed, run, w, h, spp, shadow, window)
    tr.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("VXRT_FUZZ_SEEDS", "3"))))
def test_random_frames_on_several_shares(vrt, po, gpu_device, monkeypatch, seed):
    """VORTEX_HIP_DEVICES on random inputs: 2-5 shares (the one-GPU box lists device 0 repeatedly), gathered by peer copies or through RCCL
    (every third seed), frame sizes with ragged last tile rows, fewer tile rows than shares (the run then stays on the first device), shadow
    rays, samples per pixel -- pixels and MINSTRET as on one device."""
    rng = np.random.default_rng(77000 + seed)
    sc = vrt.scene.procedural(*[("blob", 3, 0, 2), ("cornell", 0, 0, 1)][int(rng.integers(0, 2))])
    n = int(rng.integers(2, 6))
    w = int(rng.choice([8, 33, 64, 136, 200]))
    h = int(rng.choice([8, 17, 32, 61, 64, 120]))
    spp, shadow = int(rng.integers(1, 3)), bool(rng.integers(0, 2))
    monkeypatch.setenv("VORTEX_HIP_DEVICES", ",".join(["0"] * n))
    if seed % 3 == 2:
        monkeypatch.setenv("VORTEX_HIP_GATHER", "rccl")
    tr = vrt.tracer.Tracer(w, h, samples_per_pixel=spp)
    tr.init(sc)
    monkeypatch.delenv("VORTEX_HIP_DEVICES")
    monkeypatch.delenv("VORTEX_HIP_GATHER", raising=False)
    assert tr.dev.hip_stat(3) == n
    for run in range(2):
        light = (float(rng.uniform(-100, 500)), float(rng.uniform(50, 500)), float(rng.uniform(-200, 200)))
        tr.bufs["out"].write(np.full(w * h * 4, 0xA5, np.uint8))
        tr.setup(light_pos=light, shadow=shadow)
        got = tr.run()
        want, _, _, nrays = po.render_ex(sc, w, h, po.shade_params(light_pos=light), shadow=int(shadow))
        assert np.array_equal(got, want), (seed, run, n, w, h, spp, shadow)
        assert tr.dev.mpm_query(vrt.runtime.VX_CSR_MINSTRET, 0) == spp * nrays
    split = (h + 7) // 8 >= n
    assert tr.dev.hip_stat(2) == (2 if split else 0)
    if seed % 3 == 2:
        assert tr.dev.hip_stat(7) == (2 if split else 0)
    tr.close()
