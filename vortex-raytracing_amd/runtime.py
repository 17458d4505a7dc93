"""ctypes mirror of the Vortex host API (reference runtime/include/vortex.h:80-145), same names and
error convention (0 = ok, non-zero = failure).  Calls go through libvortex.so (csrc/vx_stub.cpp),
which resolves libvortex-$VORTEX_DRIVER.so -> vx_dev_init exactly like the reference dispatcher
(runtime/stub/vortex.cpp:58-82)."""
import ctypes as C
import os

from . import LIB_DIR, lib_path

VX_MEM_READ, VX_MEM_WRITE, VX_MEM_READ_WRITE = 1, 2, 3
VX_MAX_TIMEOUT = 24 * 60 * 60 * 1000
VX_DCR_BASE_RTX_TLAS_PTR, VX_DCR_BASE_RTX_BLAS_PTR, VX_DCR_BASE_RTX_BVH_PTR, VX_DCR_BASE_RTX_TRI_PTR = 6, 7, 8, 9
VX_DCR_HIP_ROW_BEGIN, VX_DCR_HIP_ROW_END, VX_DCR_HIP_SHADOW_RAYS, VX_DCR_HIP_ROW_STRIDE = 0x7F0, 0x7F1, 0x7F2, 0x7F3
VX_CAPS_NUM_THREADS, VX_CAPS_NUM_WARPS, VX_CAPS_NUM_CORES = 1, 2, 3
VX_CSR_MCYCLE, VX_CSR_MINSTRET = 0xB00, 0xB02

_lib = None
_hip = None


class VxError(RuntimeError):
    pass


def lib():
    """libvortex.so with prototypes set.  Makes sure the loader can find the backend."""
    global _lib
    if _lib is None:
        lib_path("libvortex-hip.so")  # fail loudly if the HIP backend is not built
        os.environ.setdefault("VORTEX_DRIVER", "hip")
        os.environ["LD_LIBRARY_PATH"] = LIB_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
        L = C.CDLL(lib_path("libvortex.so"))
        H, B, U64, U32, P = C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER
        sig = {
            "vx_dev_open": [P(H)], "vx_dev_close": [H], "vx_dev_caps": [H, U32, P(U64)],
            "vx_mem_alloc": [H, U64, C.c_int, P(B)], "vx_mem_reserve": [H, U64, U64, C.c_int, P(B)],
            "vx_mem_free": [B], "vx_mem_access": [B, U64, U64, C.c_int], "vx_mem_address": [B, P(U64)],
            "vx_mem_info": [H, P(U64), P(U64)], "vx_copy_to_dev": [B, C.c_void_p, U64, U64],
            "vx_copy_from_dev": [C.c_void_p, B, U64, U64], "vx_start": [H, B, B], "vx_ready_wait": [H, U64],
            "vx_dcr_read": [H, U32, P(U32)], "vx_dcr_write": [H, U32, U32], "vx_mpm_query": [H, U32, U32, P(U64)],
            "vx_upload_kernel_bytes": [H, C.c_void_p, U64, P(B)], "vx_upload_kernel_file": [H, C.c_char_p, P(B)],
            "vx_upload_bytes": [H, C.c_void_p, U64, P(B)], "vx_upload_file": [H, C.c_char_p, P(B)],
            "vx_check_occupancy": [H, U32, P(U32)], "vx_dump_perf": [H, C.c_void_p],
        }
        for name, args in sig.items():
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = args
        _lib = L
    return _lib


def hip_lib():
    """libvortex-hip.so itself (backend extension + vxrt_* direct API)."""
    global _hip
    if _hip is None:
        _hip = C.CDLL(lib_path("libvortex-hip.so"), mode=C.RTLD_GLOBAL)
        _hip.vx_hip_device_stat.restype = C.c_int
        _hip.vx_hip_device_stat.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        _hip.vx_hip_buffer_device_ptr.restype = C.c_int
        _hip.vx_hip_buffer_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    return _hip


def check(rc, what):
    if rc != 0:
        raise VxError("%s returned %d" % (what, rc))


class Buffer:
    def __init__(self, handle, size):
        self.handle = handle
        self.size = size

    @property
    def address(self):
        a = C.c_uint64()
        check(lib().vx_mem_address(self.handle, C.byref(a)), "vx_mem_address")
        return a.value

    def write(self, data, offset=0):
        mv = memoryview(data).cast("B")
        buf = (C.c_uint8 * len(mv)).from_buffer_copy(mv) if mv.readonly else (C.c_uint8 * len(mv)).from_buffer(mv)
        check(lib().vx_copy_to_dev(self.handle, buf, offset, len(mv)), "vx_copy_to_dev")

    def read(self, size=None, offset=0):
        size = self.size - offset if size is None else size
        out = bytearray(size)
        buf = (C.c_uint8 * size).from_buffer(out)
        check(lib().vx_copy_from_dev(buf, self.handle, offset, size), "vx_copy_from_dev")
        return out

    def device_ptr(self):
        p = C.c_void_p()
        check(hip_lib().vx_hip_buffer_device_ptr(self.handle, C.byref(p)), "vx_hip_buffer_device_ptr")
        return p.value

    def free(self):
        if self.handle:
            check(lib().vx_mem_free(self.handle), "vx_mem_free")
            self.handle = None


class Device:
    def __init__(self):
        h = C.c_void_p()
        check(lib().vx_dev_open(C.byref(h)), "vx_dev_open")
        self.handle = h

    def close(self):
        if self.handle:
            check(lib().vx_dev_close(self.handle), "vx_dev_close")
            self.handle = None

    def hip_stat(self, which):
        """vx_hip_device_stat: 0 = acceleration layouts built, 1 = hipMalloc calls for buffers, 2 = runs split over several GPUs, 3 = GPUs behind the device,
        4 / 5 = joined runs whose MCYCLE came from the device's clock / the host's, 6 = the last joined run on the host's clock (us), 7 = runs whose shares were gathered through RCCL (VORTEX_HIP_GATHER=rccl)"""
        v = C.c_uint64()
        check(hip_lib().vx_hip_device_stat(self.handle, which, C.byref(v)), "vx_hip_device_stat")
        return int(v.value)

    def caps(self, caps_id):
        v = C.c_uint64()
        check(lib().vx_dev_caps(self.handle, caps_id, C.byref(v)), "vx_dev_caps")
        return v.value

    def mem_alloc(self, size, flags=VX_MEM_READ_WRITE):
        b = C.c_void_p()
        check(lib().vx_mem_alloc(self.handle, size, flags, C.byref(b)), "vx_mem_alloc")
        return Buffer(b, size)

    def upload_bytes(self, data):
        mv = memoryview(data).cast("B")
        b = C.c_void_p()
        buf = (C.c_uint8 * len(mv)).from_buffer_copy(mv)
        check(lib().vx_upload_bytes(self.handle, buf, len(mv), C.byref(b)), "vx_upload_bytes")
        return Buffer(b, len(mv))

    def upload_kernel_file(self, path):
        b = C.c_void_p()
        check(lib().vx_upload_kernel_file(self.handle, str(path).encode(), C.byref(b)), "vx_upload_kernel_file")
        return Buffer(b, 0x1000)

    def dcr_write(self, addr, value):
        check(lib().vx_dcr_write(self.handle, addr, value & 0xFFFFFFFF), "vx_dcr_write")

    def dcr_read(self, addr):
        v = C.c_uint32()
        check(lib().vx_dcr_read(self.handle, addr, C.byref(v)), "vx_dcr_read")
        return v.value

    def start(self, kernel, args):
        check(lib().vx_start(self.handle, kernel.handle, args.handle), "vx_start")

    def ready_wait(self, timeout=VX_MAX_TIMEOUT):
        check(lib().vx_ready_wait(self.handle, timeout), "vx_ready_wait")

    def mpm_query(self, addr, core_id=0):
        v = C.c_uint64()
        check(lib().vx_mpm_query(self.handle, addr, core_id, C.byref(v)), "vx_mpm_query")
        return v.value
