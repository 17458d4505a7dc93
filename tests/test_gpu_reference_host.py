"""The drop-in, proven with the reference's own code: oracle/_ref/raytracing is the UNMODIFIED host program of
tests/regression/raytracing (main.cpp, tracer.cpp, its scene builder) linked against oracle/_ref/libvortex.so, the
reference's own API library + dispatcher (runtime/stub/{vortex,utils,perf}.cpp) -- both compiled where their sources
lie by oracle/ref_build/Makefile (target `host`).  With VORTEX_DRIVER=hip the dispatcher dlopens libvortex-hip.so, calls
vx_dev_init and the host program drives the HIP path through exactly the calls the reference makes: 4 x vx_upload_kernel_file
(the selector .vxbin images), 11 x vx_mem_alloc + vx_mem_address, 9 x vx_copy_to_dev, the SBT, the 4 RTX DCR writes,
vx_upload_bytes(kernel_arg_t), vx_start, vx_ready_wait, vx_copy_from_dev, vx_dev_close (-> vx_dump_perf).

The image it writes (ASCII P3, tracer.cpp:15-33) must equal the oracle's frame on the buffers the reference's scene builder
produces for the same OBJ (same builder code through oracle/_ref/libvxref.so).  Test infrastructure only."""
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HOST = os.path.join(REFDIR, "raytracing")

pytestmark = pytest.mark.gpu


def _write_png(path, rgb):
    """8-bit RGB PNG (filter 0 rows) -- the texture the reference decodes with stb_image (surface.cpp:28-55)."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(tag, data):
        c = tag + data
        return struct.pack(">I", len(data)) + c + struct.pack(">I", zlib.crc32(c) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _write_obj(vrt, d, name):
    """A textured blob and a plain floor in front of the RTU kernel's fixed camera, as OBJ + MTL + PNG."""
    blob = vrt.scene.procedural("blob", 3, 0, 5)
    tri = blob["tri"].view(np.float32).reshape(-1, 3, 3)
    tex = blob["triEx"].view(np.float32).reshape(-1, 16)
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 256, size=(23, 37, 3))).astype(np.uint8)      # odd sizes on purpose
    img[::4] = (250, 40, 40)
    _write_png(os.path.join(d, "checker.png"), img)
    with open(os.path.join(d, "scene.mtl"), "w") as f:
        f.write("newmtl skin\nKa 0.1 0.1 0.1\nKd 0.8 0.8 0.8\nmap_Kd checker.png\n\nnewmtl floor\nKa 0.1 0.1 0.1\nKd 0.35 0.55 0.4\n")
    with open(os.path.join(d, name), "w") as f:
        f.write("mtllib scene.mtl\n")
        n = 0
        f.write("usemtl skin\n")
        for t, e in zip(tri, tex):
            for k in range(3):
                f.write("v %.9g %.9g %.9g\n" % tuple(t[k]))
                f.write("vn %.9g %.9g %.9g\n" % tuple(e[3 * k: 3 * k + 3]))
                f.write("vt %.9g %.9g\n" % (e[9 + 2 * k] * 1.5 - 2.25, e[10 + 2 * k] * 1.5 - 0.5))   # uv on both sides of [0,1]
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % tuple(n + i for i in (1, 1, 1, 2, 2, 2, 3, 3, 3)))
            n += 3
        f.write("usemtl floor\n")
        quad = [(60, 20, -160), (420, 20, -160), (420, 20, 160), (60, 20, 160)]
        for v in quad:
            f.write("v %g %g %g\nvn 0 1 0\nvt 0 0\n" % v)
        f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % tuple(n + i for i in (1, 1, 1, 2, 2, 2, 3, 3, 3)))
        f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % tuple(n + i for i in (1, 1, 1, 3, 3, 3, 4, 4, 4)))


def _read_ppm(path):
    """Inverse of tracer.cpp:15-33: rows bottom-up, each pixel written as bytes 2, 1, 0 of the little-endian u32."""
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    v = np.array(tok[4:], np.float64).astype(np.uint32).reshape(h, w, 3)
    px = (v[..., 0] << 16) | (v[..., 1] << 8) | v[..., 2]
    return px[::-1].copy()


def test_reference_host_program_renders_through_the_hip_backend(vrt, po, gpu_device, tmp_path):
    if not os.path.exists(HOST) or not os.path.exists(os.path.join(REFDIR, "libvortex.so")):
        pytest.fail("oracle/_ref/raytracing is not built (make -C oracle/ref_build host, where /root/reference exists)")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "assets"))
    _write_obj(vrt, os.path.join(d, "assets"), "scene.obj")
    for k in ("kernel", "miss", "closest", "anyhit"):     # the host program opens them in its working directory (tracer.cpp:117-120)
        shutil.copy(os.path.join(vrt.VXBIN_DIR, k + ".vxbin"), os.path.join(d, k + ".vxbin"))
    w, h = 200, 120
    env = dict(os.environ, VORTEX_DRIVER="hip",
               LD_LIBRARY_PATH=REFDIR + os.pathsep + vrt.LIB_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([HOST, "-m", "scene.obj", "-w", str(w), "-h", str(h), "-o", "out.ppm"], cwd=d, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "Image saved to" in r.stdout
    got = _read_ppm(os.path.join(d, "out.ppm"))
    # the same scene through the reference's builder in this process, rendered by the oracle
    sc = po.ref_scene([os.path.join(d, "assets", "scene.obj")])
    pp = po.shade_params()      # the host program's defaults (main.cpp:34-41)
    want, hits, _ = po.render(sc, w, h, pp)
    hit = hits["dist"] < 1e29
    assert hit.mean() > 0.2 and (~hit).any()
    te = sc["triEx"].view(np.float32).reshape(-1, 16)
    mat_of_hit = te[hits["triIdx"][hit], 15].view(np.int32)
    textured = sc["mat"].view(np.int32).reshape(-1, 22)[mat_of_hit, 16] >= 0
    assert textured.any() and (~textured).any()
    np.testing.assert_array_equal(got, want)
    # vx_dev_close ran the reference's own vx_dump_perf (runtime/stub/perf.cpp:195-227,557) against the backend's dev_caps /
    # mpm_query answers: its summary line carries the rays of the run (MINSTRET, core 0) and a positive cycle count (MCYCLE)
    import re
    m = re.search(r"PERF: instrs=(\d+), cycles=(\d+), IPC=", r.stdout)
    assert m, r.stdout[-1500:]
    assert int(m.group(1)) == w * h and int(m.group(2)) > 0      # one primary ray per pixel (the reference host renders without the shadow extension)
