"""ctypes binding of the vxrt_* direct launch API (include/vortex_hip.h level 2): launches on
caller-owned DEVICE pointers and a caller-owned HIP stream.  Used by bench.py and the GPU tests with
torch tensors as plain device memory (`tensor.data_ptr()`), nothing torch-typed crosses the ABI."""
import ctypes as C

from .runtime import hip_lib, check

MODE_CLOSEST, MODE_ANY = 0, 1


class VxrtScene(C.Structure):
    _fields_ = [("tlas", C.c_void_p), ("blas", C.c_void_p), ("bvh", C.c_void_p), ("tri", C.c_void_p),
                ("triEx", C.c_void_p), ("mat", C.c_void_p), ("tex", C.c_void_p),
                ("n_tlas_nodes", C.c_uint32), ("n_blas", C.c_uint32), ("n_bvh_nodes", C.c_uint32),
                ("n_tris", C.c_uint32), ("n_mats", C.c_uint32), ("reserved", C.c_uint32), ("tex_bytes", C.c_uint64)]


class ShadeParams(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("light_color", C.c_float * 3), ("light_pos", C.c_float * 3),
                ("background", C.c_float * 3), ("max_depth", C.c_uint32)]


# defaults of the reference host program (tests/regression/raytracing/main.cpp:34-41)
def default_shade_params():
    p = ShadeParams()
    p.ambient[:] = (0.4, 0.4, 0.4)
    p.light_color[:] = (1.0, 1.0, 1.0)
    p.light_pos[:] = (0.0, 10.0, -10.0)
    p.background[:] = (0.4, 0.35, 0.25)
    p.max_depth = 1
    return p


_proto = False


def _lib():
    global _proto
    L = hip_lib()
    if not _proto:
        L.vxrt_accel_build.restype = C.c_int
        L.vxrt_accel_build.argtypes = [C.POINTER(VxrtScene), C.c_void_p, C.POINTER(C.c_void_p)]
        L.vxrt_accel_destroy.restype = C.c_int
        L.vxrt_accel_destroy.argtypes = [C.c_void_p]
        L.vxrt_accel_bytes.restype = C.c_uint64
        L.vxrt_accel_bytes.argtypes = [C.c_void_p]
        L.vxrt_accel_frames_in_flight.restype = C.c_int
        L.vxrt_accel_frames_in_flight.argtypes = [C.c_void_p, C.c_uint32]
        L.vxrt_render.restype = C.c_int
        L.vxrt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.POINTER(ShadeParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vxrt_render_stats.restype = C.c_int
        L.vxrt_render_ao.restype = C.c_int
        L.vxrt_render_ao.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams),
                                     C.POINTER(AoParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vxrt_render_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(ShadeParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vxrt_trace.restype = C.c_int
        L.vxrt_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.vxrt_status.restype = C.c_int
        L.vxrt_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.vxrt_version.restype = C.c_char_p
        _proto = True
    return L


def accel_build(scene, stream=None):
    """Build the device-side acceleration layout for a VxrtScene; returns the opaque handle."""
    h = C.c_void_p()
    check(_lib().vxrt_accel_build(C.byref(scene), stream, C.byref(h)), "vxrt_accel_build")
    return h


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("max_leaf", C.c_uint32), ("max_depth", C.c_uint32),
                ("bounds", C.c_float * 6)]


class BvhTooDeep(RuntimeError):
    """vxrt_bvh_build: the Morton-order tree is deeper than the 32 levels the reference's trail supports."""


def bvh_build(tri_ptr, triEx_ptr, n_tris, nodes_ptr, node_capacity, tri_offset=0, leaf_max=0, stream=None):
    """vxrt_bvh_build: BLAS of one mesh on the GPU, in the reference's node format; tri / triEx are reordered in place."""
    L = _lib()
    L.vxrt_bvh_build.restype = C.c_int
    L.vxrt_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(BvhInfo), C.c_void_p]
    info = BvhInfo()
    rc = L.vxrt_bvh_build(tri_ptr, triEx_ptr, n_tris, tri_offset, leaf_max, nodes_ptr, node_capacity, C.byref(info), stream)
    if rc == -2:
        raise BvhTooDeep("vxrt_bvh_build: tree depth %d exceeds the reference's 32 levels" % info.max_depth)
    check(rc, "vxrt_bvh_build")
    return info


def tlas_build(boxes_ptr, n_instances, nodes_ptr, node_capacity, stream=None):
    """vxrt_tlas_build: TLAS over n_instances world-space boxes (device, 6 floats each), in the reference's node format."""
    L = _lib()
    L.vxrt_tlas_build.restype = C.c_int
    L.vxrt_tlas_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(BvhInfo), C.c_void_p]
    info = BvhInfo()
    rc = L.vxrt_tlas_build(boxes_ptr, n_instances, nodes_ptr, node_capacity, C.byref(info), stream)
    if rc == -2:
        raise BvhTooDeep("vxrt_tlas_build: tree depth %d exceeds the reference's 32 levels" % info.max_depth)
    check(rc, "vxrt_tlas_build")
    return info


def accel_destroy(accel):
    if accel:
        check(_lib().vxrt_accel_destroy(accel), "vxrt_accel_destroy")


def accel_frames_in_flight(accel, n):
    """vxrt_accel_frames_in_flight: frames this accel keeps in flight (1..8); raises on a bad count."""
    if _lib().vxrt_accel_frames_in_flight(accel, int(n)) != 0:
        raise RuntimeError("vxrt_accel_frames_in_flight(%r) failed" % (n,))


def trace_reference_quirks(image_ptr, image_size, offsets, rays_ptr, n, hits_ptr, mode=MODE_CLOSEST, tmax_ptr=None, stream=None):
    """vxrt_trace_reference_quirks: the RTU's traversal restated literally (stale base_ptr included) on a flat memory image;
    offsets = (tlas, blas, bvh, tri) byte offsets into the image, i.e. the values of the RTX DCRs 0x6..0x9."""
    L = _lib()
    L.vxrt_trace_reference_quirks.restype = C.c_int
    L.vxrt_trace_reference_quirks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    check(L.vxrt_trace_reference_quirks(image_ptr, image_size, offsets[0], offsets[1], offsets[2], offsets[3], rays_ptr, n, tmax_ptr, hits_ptr, mode, stream),
          "vxrt_trace_reference_quirks")


def debug_read_control(accel, ctx=0, n_dwords=800, stream=None):
    """vxrt_debug_read_control: the control block of frame context `ctx` (numpy u32) as the last call left it."""
    import numpy as np
    L = _lib()
    L.vxrt_debug_read_control.restype = C.c_int
    L.vxrt_debug_read_control.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    out = np.zeros(n_dwords, np.uint32)
    check(L.vxrt_debug_read_control(accel, ctx, out.ctypes.data, n_dwords, stream), "vxrt_debug_read_control")
    return out


def accel_bytes(accel):
    return int(_lib().vxrt_accel_bytes(accel))


def render(accel, width, height, y0, y1, params, dst_ptr, shadow=0, hits_ptr=None, colors_ptr=None, rays_ptr=None, stream=None):
    check(_lib().vxrt_render(accel, width, height, y0, y1, C.byref(params), int(shadow), dst_ptr, hits_ptr,
                             colors_ptr, rays_ptr, stream), "vxrt_render")


def render_interleaved(accel, width, height, phase, stride, params, dst_ptr, shadow=0, hits_ptr=None, colors_ptr=None, rays_ptr=None, stream=None):
    """vxrt_render_interleaved: the tile rows phase, phase + stride, ... of the frame (one rank's share of a frame split over GPUs)."""
    L = _lib()
    L.vxrt_render_interleaved.restype = C.c_int
    L.vxrt_render_interleaved.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.vxrt_render_interleaved(accel, width, height, int(phase), int(stride), C.byref(params), int(shadow), dst_ptr, hits_ptr,
                                    colors_ptr, rays_ptr, stream), "vxrt_render_interleaved")


def trace_stats(accel, rays_ptr, n, hits_ptr, mode=0, tmax_ptr=None, stream=None):
    """vxrt_trace_stats: counting build of the ray-buffer traversal; returns counts and SURVEY s8d bytes per ray
    (24 B ray read + 52 B per node / instance record + 36 B per triangle + 24 B hit record written)."""
    import torch
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda:%d" % torch.cuda.current_device())
    L = _lib()
    L.vxrt_trace_stats.restype = C.c_int
    L.vxrt_trace_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    check(L.vxrt_trace_stats(accel, rays_ptr, n, tmax_ptr, hits_ptr, mode, cnt.data_ptr(), stream), "vxrt_trace_stats")
    torch.cuda.synchronize()
    r, nn, ni, nt = (int(v) for v in cnt.tolist()[:4])
    total = 24 * r + 52 * (nn + ni) + 36 * nt + 24 * r
    return {"rays": r, "node_fetches": nn, "inst_fetches": ni, "tri_fetches": nt, "bytes": total, "bytes_per_ray": total / max(r, 1)}


class RcScene(C.Structure):    # vxrc_scene_t
    _fields_ = [(k, C.c_void_p) for k in ("tlas", "blas", "bvh", "tri", "triEx", "triIdx", "tex")] + \
               [(k, C.c_uint32) for k in ("n_tlas_nodes", "n_blas", "n_bvh_nodes", "n_tris", "n_tri_idx", "tlas_root")] + \
               [("tex_bytes", C.c_uint64)]


class RcParams(C.Structure):   # vxrc_params_t
    _fields_ = [("camera_pos", C.c_float * 3), ("camera_forward", C.c_float * 3), ("camera_right", C.c_float * 3),
                ("camera_up", C.c_float * 3), ("viewplane", C.c_float * 2),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("light_pos", C.c_float * 3), ("light_color", C.c_float * 3), ("ambient_color", C.c_float * 3),
                ("background_color", C.c_float * 3)]


def rc_params(cam14, light12, spp=1, max_depth=1):
    p = RcParams()
    c = [float(v) for v in cam14]
    p.camera_pos[:] = c[0:3]; p.camera_forward[:] = c[3:6]; p.camera_right[:] = c[6:9]; p.camera_up[:] = c[9:12]; p.viewplane[:] = c[12:14]
    l = [float(v) for v in light12]
    p.light_pos[:] = l[0:3]; p.light_color[:] = l[3:6]; p.ambient_color[:] = l[6:9]; p.background_color[:] = l[9:12]
    p.samples_per_pixel, p.max_depth = int(spp), int(max_depth)
    return p


def rc_render(scene, width, height, y0, y1, params, dst_ptr, colors_ptr=None, stream=None):
    """vxrc_render: the software twin (tests/regression/raycast) on device buffers described by an RcScene."""
    L = _lib()
    L.vxrc_render.restype = C.c_int
    L.vxrc_render.argtypes = [C.POINTER(RcScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RcParams),
                              C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.vxrc_render(C.byref(scene), width, height, y0, y1, C.byref(params), dst_ptr, colors_ptr, stream), "vxrc_render")


def rc_accel_build(scene, stream=None):
    """vxrc_accel_build: the twin's acceleration layout for an RcScene; returns the opaque handle."""
    L = _lib()
    L.vxrc_accel_build.restype = C.c_int
    L.vxrc_accel_build.argtypes = [C.POINTER(RcScene), C.c_void_p, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    check(L.vxrc_accel_build(C.byref(scene), stream, C.byref(h)), "vxrc_accel_build")
    return h


def wire_pack(frames_ptr, frame_stride, width, tile_rows_per_rank, world, rank, n_frames, wire_ptr, stream=None):
    """vxrt_wire_pack: this rank's interleaved tile rows of n_frames frames as 3 bytes per pixel."""
    L = _lib()
    L.vxrt_wire_pack.restype = C.c_int
    L.vxrt_wire_pack.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    check(L.vxrt_wire_pack(frames_ptr, frame_stride, width, tile_rows_per_rank, world, rank, n_frames, wire_ptr, stream), "vxrt_wire_pack")


def wire_unpack(wire_all_ptr, wire_stride_bytes, width, tile_rows_per_rank, world, n_frames, frames_ptr, frame_stride, stream=None):
    """vxrt_wire_unpack: the `world` shares expanded and interleaved into n_frames frames of padded height."""
    L = _lib()
    L.vxrt_wire_unpack.restype = C.c_int
    L.vxrt_wire_unpack.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    check(L.vxrt_wire_unpack(wire_all_ptr, wire_stride_bytes, width, tile_rows_per_rank, world, n_frames, frames_ptr, frame_stride, stream), "vxrt_wire_unpack")


def accel_info(accel, which):
    """vxrt_accel_info: 0 -> internal levels of the deepest path (counted up to 17), 1 -> 48-entry stacks (depth class <= 16),
    2 -> single identity instance under the TLAS root, 3 -> ldexp decode."""
    L = _lib()
    L.vxrt_accel_info.restype = C.c_int
    L.vxrt_accel_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    v = C.c_uint64()
    check(L.vxrt_accel_info(accel, which, C.byref(v)), "vxrt_accel_info")
    return int(v.value)


def rc_accel_info(accel, which):
    """vxrc_accel_info: 0 -> the walk takes two levels per fetch (wide nodes), 1 -> internal levels of the deepest tree."""
    L = _lib()
    L.vxrc_accel_info.restype = C.c_int
    L.vxrc_accel_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    v = C.c_uint64()
    check(L.vxrc_accel_info(accel, which, C.byref(v)), "vxrc_accel_info")
    return int(v.value)


def rc_accel_destroy(accel):
    if accel:
        L = _lib()
        L.vxrc_accel_destroy.restype = C.c_int
        L.vxrc_accel_destroy.argtypes = [C.c_void_p]
        check(L.vxrc_accel_destroy(accel), "vxrc_accel_destroy")


def rc_render_accel(accel, width, height, y0, y1, params, dst_ptr, colors_ptr=None, stream=None):
    """vxrc_render_accel: the software twin on a prebuilt layout."""
    L = _lib()
    L.vxrc_render_accel.restype = C.c_int
    L.vxrc_render_accel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RcParams), C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.vxrc_render_accel(accel, width, height, y0, y1, C.byref(params), dst_ptr, colors_ptr, stream), "vxrc_render_accel")


class AoParams(C.Structure):   # vxrt_ao_params_t
    _fields_ = [("spp", C.c_uint32), ("radius", C.c_float), ("seed", C.c_uint32), ("reserved", C.c_uint32)]


def render_ao(accel, width, height, y0, y1, params, spp, radius, dst_ptr, seed=0, colors_ptr=None, unoccluded_ptr=None,
              rays_ptr=None, stream=None):
    """vxrt_render_ao: primary hit -> Lambert colour x fraction of `spp` occlusion rays (tmax = radius) that reach nothing."""
    ao = AoParams(int(spp), float(radius), int(seed), 0)
    check(_lib().vxrt_render_ao(accel, width, height, y0, y1, C.byref(params), C.byref(ao), dst_ptr, colors_ptr,
                                unoccluded_ptr, rays_ptr, stream), "vxrt_render_ao")


def render_diffuse_bounce(accel, width, height, y0, y1, params, dst_ptr, seed=0, colors_ptr=None, rays_ptr=None, stream=None):
    """vxrt_render_diffuse_bounce: primary hit + one cosine-weighted closest-hit bounce."""
    L = _lib()
    L.vxrt_render_diffuse_bounce.restype = C.c_int
    L.vxrt_render_diffuse_bounce.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.vxrt_render_diffuse_bounce(accel, width, height, y0, y1, C.byref(params), int(seed), dst_ptr, colors_ptr, rays_ptr, stream),
          "vxrt_render_diffuse_bounce")


STAT_KEYS = ("rays", "node_fetches", "inst_fetches", "tri_fetches", "shaded_hits", "textured_hits", "pixels")


def algorithmic_bytes(c):
    """SURVEY.md s8d per-ray formula summed over one launch of the render kernel: 52 B per node and
    per instance record fetched, 36 B per triangle tested, 64 B triEx + 88 B material (+4 B texel)
    per shaded hit, 4 B per pixel written.  Rays are generated in registers and hit records stay in
    registers, so the formula's 24 B ray read / 24 B hit write do not apply to this kernel."""
    return (52 * (c["node_fetches"] + c["inst_fetches"]) + 36 * c["tri_fetches"]
            + (64 + 88) * c["shaded_hits"] + 4 * c["textured_hits"] + 4 * c["pixels"])


def render_interleaved_batch(accel, width, height, phase, stride, params_list, dst_ptr, dst_frame_stride, shadow=0, rays_ptr=None, stream=None):
    """vxrt_render_interleaved_batch: len(params_list) frames of this share in one set of launches; frame f -> dst + f * dst_frame_stride pixels."""
    L = _lib()
    L.vxrt_render_interleaved_batch.restype = C.c_int
    L.vxrt_render_interleaved_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_int,
                                                C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    arr = (ShadeParams * len(params_list))(*params_list)
    check(L.vxrt_render_interleaved_batch(accel, width, height, phase, stride, len(params_list), arr, int(shadow), dst_ptr, dst_frame_stride, rays_ptr, stream),
          "vxrt_render_interleaved_batch")


def render_rows_batch(accel, width, height, y0, y1, params_list, dst_ptr, dst_frame_stride, shadow=0, rays_ptr=None, stream=None):
    """vxrt_render_rows_batch: len(params_list) frames of the row band [y0, y1) in one set of launches.  dst_ptr addresses a full
    frame (pixel (x, y) of frame f at dst + f * dst_frame_stride + x + y * width, in pixels): a caller that keeps only its band
    passes band_ptr - 4 * y0 * width and a frame stride of (y1 - y0) * width."""
    L = _lib()
    L.vxrt_render_rows_batch.restype = C.c_int
    L.vxrt_render_rows_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_int,
                                         C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    arr = (ShadeParams * len(params_list))(*params_list)
    check(L.vxrt_render_rows_batch(accel, width, height, y0, y1, len(params_list), arr, int(shadow), dst_ptr, dst_frame_stride, rays_ptr, stream),
          "vxrt_render_rows_batch")


def render_batch(accel, width, height, params_list, dst_ptr, dst_frame_stride, shadow=0, rays_ptr=None, stream=None):
    """vxrt_render_batch: len(params_list) whole frames in one set of launches; frame f -> dst + f * dst_frame_stride pixels."""
    L = _lib()
    L.vxrt_render_batch.restype = C.c_int
    L.vxrt_render_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    arr = (ShadeParams * len(params_list))(*params_list)
    check(L.vxrt_render_batch(accel, width, height, len(params_list), arr, int(shadow), dst_ptr, dst_frame_stride, rays_ptr, stream), "vxrt_render_batch")


def render_stats(accel, width, height, y0, y1, params, dst_ptr, shadow=0, stream=None, timed=False):
    """Runs the counting build of the render kernels once; returns the counters + algorithmic bytes.  timed=True counts the
    traversal the timed kernel performs (unordered occlusion rays, leaf helpers) instead of the reference-order one."""
    import torch
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda:%d" % torch.cuda.current_device())
    L = _lib()
    fn = L.vxrt_render_stats_timed if timed else L.vxrt_render_stats
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShadeParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    check(fn(accel, width, height, y0, y1, C.byref(params), int(shadow), dst_ptr, cnt.data_ptr(), stream), "vxrt_render_stats")
    torch.cuda.synchronize()
    c = dict(zip(STAT_KEYS, [int(v) for v in cnt[:7].tolist()]))
    c["bytes"] = algorithmic_bytes(c)
    return c


def trace(accel, rays_ptr, n, hits_ptr, mode=MODE_CLOSEST, tmax_ptr=None, stream=None):
    check(_lib().vxrt_trace(accel, rays_ptr, n, tmax_ptr, hits_ptr, mode, stream), "vxrt_trace")


def shade_rays(accel, rays_ptr, hits_ptr, n, params, colors_ptr=None, rgb8_ptr=None, stream=None):
    """vxrt_shade_rays: closest-hit / miss shader over (ray, hit record) pairs."""
    L = _lib()
    L.vxrt_shade_rays.restype = C.c_int
    L.vxrt_shade_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(ShadeParams), C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.vxrt_shade_rays(accel, rays_ptr, hits_ptr, n, C.byref(params), colors_ptr, rgb8_ptr, stream), "vxrt_shade_rays")


def status(stream=None):
    st = C.c_uint32()
    check(_lib().vxrt_status(stream, C.byref(st)), "vxrt_status")
    return st.value


def version():
    return _lib().vxrt_version().decode()
