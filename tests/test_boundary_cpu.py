"""CPU (no GPU): the C-ABI library loads and exports every symbol include/vortex_hip.h declares, the
plug-in entry fills all 16 callbacks, the kernel-selector images have the reference's vxbin layout,
and the host-side structures have the reference's sizes.  No compute call is made here."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(vrt):
    hdr = open(os.path.join(ROOT, "include", "vortex_hip.h")).read()
    lib = C.CDLL(vrt.lib_path("libvortex-hip.so"))
    declared = set(re.findall(r"^(?:int|const char\*)\s+(vx\w+|vxrt_\w+)\s*\(", hdr, re.M))
    assert {"vx_dev_init", "vxrt_render", "vxrt_render_stats", "vxrt_trace", "vxrt_status", "vxrt_version", "vx_hip_buffer_device_ptr"} <= declared
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_vx_dev_init_fills_all_sixteen_callbacks(vrt):
    lib = C.CDLL(vrt.lib_path("libvortex-hip.so"))
    cb = (C.c_void_p * 16)()
    lib.vx_dev_init.argtypes = [C.c_void_p]
    assert lib.vx_dev_init(cb) == 0
    assert all(cb[i] for i in range(16))      # callbacks.h:23-72 has exactly 16 members
    assert lib.vx_dev_init(None) != 0          # callbacks.inc:21-22


def test_host_api_library_exports_the_reference_api(vrt):
    lib = C.CDLL(vrt.lib_path("libvortex.so"))
    api = ["vx_dev_open", "vx_dev_close", "vx_dev_caps", "vx_mem_alloc", "vx_mem_reserve", "vx_mem_free", "vx_mem_access",
           "vx_mem_address", "vx_mem_info", "vx_copy_to_dev", "vx_copy_from_dev", "vx_start", "vx_ready_wait", "vx_dcr_read",
           "vx_dcr_write", "vx_mpm_query", "vx_upload_kernel_bytes", "vx_upload_kernel_file", "vx_upload_bytes", "vx_upload_file",
           "vx_check_occupancy", "vx_dump_perf"]   # runtime/include/vortex.h:80-145 (17 + utilities)
    for f in api:
        assert hasattr(lib, f), f


def test_kernel_selector_images_follow_the_vxbin_layout(vrt):
    vmas = {"kernel": 0x80000000, "miss": 0x80100000, "closest": 0x80200000, "anyhit": 0x80300000}  # raytracing/Makefile:104-107
    for name, vma in vmas.items():
        blob = open(os.path.join(vrt.VXBIN_DIR, name + ".vxbin"), "rb").read()
        lo, hi = struct.unpack_from("<QQ", blob)      # kernel/scripts/vxbin.py:53-74
        assert lo == vma and hi > lo and len(blob) - 16 <= hi - lo
        assert blob[16:].startswith(b"VXHIP1:raytracing." + name.encode())


def test_structure_sizes_match_the_reference(vrt):
    assert struct.calcsize(vrt.tracer.KERNEL_ARG_FMT) == 216      # raytracing/common.h:164-195
    assert C.sizeof(vrt.rtapi.VxrtScene) == 7 * 8 + 6 * 4 + 8
    sc = vrt.scene.procedural("cornell")
    assert sc["tlas"].size % 52 == 0 and sc["bvh"].size % 52 == 0 and sc["blas"].size % 160 == 0
    assert sc["tri"].size == 12 * 36 and sc["triEx"].size == 12 * 64 and sc["mat"].size % 88 == 0
    hits = np.zeros(1, dtype=[("dist", "<f4"), ("b", "<f4", 3), ("blas", "<u4"), ("tri", "<u4")])
    assert hits.itemsize == 24


def test_compute_entry_points_fail_loudly_without_a_device(vrt):
    """No CPU fallback: on a box without a HIP device opening the device must fail (non-zero), not
    silently route elsewhere.  (On the GPU box this test is a no-op.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vrt.runtime.VxError):
        vrt.runtime.Device()


def test_missing_library_raises(vrt, monkeypatch):
    monkeypatch.setattr(vrt, "LIB_DIR", "/nonexistent")
    with pytest.raises(RuntimeError):
        vrt.lib_path("libvortex-hip.so")
