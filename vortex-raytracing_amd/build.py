"""Build every native artefact of the package in-tree (hipcc cross-compiles gfx950 without a GPU).

  lib/libvortex-hip.so   HIP kernels + vx_dev_init backend + vxrt_* direct API   (hipcc, gfx950)
  lib/libvxrt_calib.so   measurement only: VALU calibration loops + clock probe  (hipcc, gfx950; not part of the product library)
  lib/libvortex.so       vx_* host API dispatcher (stand-in for the reference's runtime/stub)
  lib/libvxrt_scene.so   BVH4 builder / quantiser / procedural scenes            (g++)
  lib/rt_host            C++ host program mirroring tests/regression/raytracing/main.cpp
  vxbin/*.vxbin          kernel-selector images (16-byte vxbin header + "VXHIP1:<name>")

Run as `python vortex-raytracing_amd/build.py` or through __graft_entry__.build().
"""
import os
import shutil
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
VXBIN = os.path.join(HERE, "vxbin")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

# -ffp-contract=off is part of the numerical contract of the hit path
# -fno-slp-vectorize: under plain -O3 the SLP vectoriser packs the slab arithmetic into v_pk_mul/add_f32 + v_mov shuffles, which
# costs registers and 3 % of the frame (profiles/r02_a_variants.txt)
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
HIP_FLAGS += os.environ.get("VXRT_EXTRA_HIPFLAGS", "").split()   # experiments only (e.g. -DLDS_STACK=8)
CXX_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall"]

PRODUCT_HIP_SOURCES = ("rt_kernels.hip", "rc_kernels.hip", "vx_backend.hip", "bvh_builder.hip")

# fixed VMAs of the four images of the RTU test (tests/regression/raytracing/Makefile:104-107)
SELECTORS = {
    "kernel": (0x80000000, "raytracing.kernel"),
    "miss": (0x80100000, "raytracing.miss"),
    "closest": (0x80200000, "raytracing.closest"),
    "anyhit": (0x80300000, "raytracing.anyhit"),
    # the software twin's single kernel image (tests/regression/raycast loads "kernel.vxbin" from its own directory)
    "raycast/kernel": (0x80000000, "raycast.kernel"),
}


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def write_selectors():
    os.makedirs(VXBIN, exist_ok=True)
    for name, (vma, tag) in SELECTORS.items():
        payload = ("VXHIP1:" + tag).encode() + b"\0"
        payload += b"\0" * (64 - len(payload))
        blob = struct.pack("<QQ", vma, vma + 0x1000) + payload   # kernel/scripts/vxbin.py:53-74
        os.makedirs(os.path.dirname(os.path.join(VXBIN, name)), exist_ok=True)
        with open(os.path.join(VXBIN, name + ".vxbin"), "wb") as f:
            f.write(blob)


# The two LDS-staging variants north_star prescribes (top of the tree, triangles of shared leaves) lost their A/Bs and are compiled out of the
# shipped library; tests/test_gpu_variants.py runs the parity tests on a library with BOTH switched on so that the code cannot rot.  Built
# here (hipcc cross-compiles without a GPU) so that the GPU box loads it instead of compiling it.
TEST_VARIANT_FLAGS = ["-DRT_TOP_NODES=84", "-DRT_LDS_STACK_RENDER=5", "-DRT_LDS_STACK_RENDER_PACKED=4", "-DRT_TRI_LDS=4", "-DRT_TRI_LDS_MIN=2"]


def build_test_variant(force=False):
    d = os.path.join(HERE, "lib_ab", "lds_staging")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "libvortex-hip.so")
    src = [os.path.join(CSRC, f) for f in PRODUCT_HIP_SOURCES]
    hdrs = [os.path.join(CSRC, "rt_types.h"), os.path.join(HERE, "..", "include", "vortex_hip.h")]
    if force or _newer(so, src + hdrs):
        _run([HIPCC] + HIP_FLAGS + TEST_VARIANT_FLAGS + ["-shared", "-o", so] + src)
    for f in ("libvortex.so", "libvxrt_scene.so"):
        shutil.copy2(os.path.join(LIB, f), os.path.join(d, f))
    return d


def build(force=False, verbose=True):
    os.makedirs(LIB, exist_ok=True)
    hdrs = [os.path.join(CSRC, "rt_types.h"), os.path.join(HERE, "..", "include", "vortex_hip.h")]
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: the HIP path cannot be built" % HIPCC)

    hip_so = os.path.join(LIB, "libvortex-hip.so")
    hip_src = [os.path.join(CSRC, f) for f in PRODUCT_HIP_SOURCES]
    if force or _newer(hip_so, hip_src + hdrs):
        _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", hip_so] + hip_src)

    # measurement only (VALU calibration loops, clock probe): its own library, not part of the product
    cal_so = os.path.join(LIB, "libvxrt_calib.so")
    cal_src = [os.path.join(CSRC, "calib_kernels.hip")]
    if force or _newer(cal_so, cal_src):
        _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", cal_so] + cal_src)

    stub_so = os.path.join(LIB, "libvortex.so")
    stub_src = [os.path.join(CSRC, "vx_stub.cpp")]
    if force or _newer(stub_so, stub_src + hdrs):
        _run(["g++"] + CXX_FLAGS + ["-shared", "-o", stub_so] + stub_src + ["-ldl"])

    scene_so = os.path.join(LIB, "libvxrt_scene.so")
    scene_src = [os.path.join(CSRC, "scene_builder.cpp")]
    if force or _newer(scene_so, scene_src + hdrs):
        _run(["g++"] + CXX_FLAGS + ["-pthread", "-shared", "-o", scene_so] + scene_src + ["-lz"])

    host = os.path.join(LIB, "rt_host")
    host_src = [os.path.join(CSRC, "rt_host.cpp")]
    if os.path.exists(host_src[0]) and (force or _newer(host, host_src + hdrs + [stub_so, scene_so])):
        _run(["g++"] + CXX_FLAGS + ["-o", host] + host_src + ["-L" + LIB, "-lvortex", "-lvxrt_scene", "-pthread", "-Wl,-rpath,$ORIGIN"])

    write_selectors()
    build_test_variant(force)
    return {"hip": hip_so, "stub": stub_so, "scene": scene_so}


if __name__ == "__main__":
    build(force="--force" in sys.argv)
