"""ctypes loader of the checker libraries.  TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "librt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvxref.so")

HIT_DTYPE = np.dtype([("dist", "<f4"), ("bx", "<f4"), ("by", "<f4"), ("bz", "<f4"), ("blasIdx", "<u4"), ("triIdx", "<u4")])
assert HIT_DTYPE.itemsize == 24

STAT_FIELDS = ("node_reads", "inst_reads", "tri_reads", "accepts", "restarts", "stale_base", "trail_overflow", "abandon", "oob", "max_stack")


class OrcStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in STAT_FIELDS]

    def as_dict(self):
        return {k: getattr(self, k) for k in STAT_FIELDS}


class RefStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("node_reads", "inst_reads", "tri_reads", "bytes", "accepts", "oob")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ShadeParams(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("light_color", C.c_float * 3), ("light_pos", C.c_float * 3),
                ("background", C.c_float * 3), ("max_depth", C.c_uint32)]


def shade_params(ambient=(0.4, 0.4, 0.4), light_color=(1, 1, 1), light_pos=(0, 10, -10), background=(0.4, 0.35, 0.25), max_depth=1):
    p = ShadeParams()
    p.ambient[:] = ambient
    p.light_color[:] = light_color
    p.light_pos[:] = light_pos
    p.background[:] = background
    p.max_depth = max_depth
    return p


_orc = None
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def orc():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO):
            from . import build as _b
            _b.build()
        L = C.CDLL(ORACLE_SO)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.orc_trace_faithful.restype = C.c_int
        L.orc_trace_faithful.argtypes = [vp, u64, u32, u32, u32, u32, vp, u64, vp, vp, C.POINTER(OrcStats), C.c_int]
        L.orc_trace_canonical.restype = C.c_int
        L.orc_trace_canonical.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp, C.POINTER(OrcStats), C.c_int]
        L.orc_generate_ray.argtypes = [u32, u32, u32, u32, vp]
        L.orc_shade.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(ShadeParams), vp]
        L.orc_pack_rgb8.restype = u32
        L.orc_pack_rgb8.argtypes = [vp]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ShadeParams), vp, vp, vp]
        L.orc_render_ao.restype = C.c_int
        L.orc_render_ao.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ShadeParams), u32, C.c_float, u32,
                                    vp, vp, vp, C.POINTER(C.c_uint64)]
        L.orc_render_ex.restype = C.c_int
        L.orc_render_ex.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(ShadeParams), C.c_int, vp, vp, vp,
                                    C.POINTER(C.c_uint64)]
        L.orc_ray_box.restype = C.c_float
        L.orc_ray_box.argtypes = [vp] + [C.c_float] * 6
        _orc = L
    return _orc


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.vxref_trace.restype = C.c_int
        L.vxref_trace.argtypes = [vp, u64, u32, u32, u32, u32, vp, u64, vp, C.POINTER(RefStats), C.c_int]
        L.vxref_scene_create.restype = vp
        L.vxref_scene_create.argtypes = [C.POINTER(C.c_char_p), C.c_int]
        L.vxref_scene_destroy.argtypes = [vp]
        L.vxref_scene_buffer.restype = u64
        L.vxref_scene_buffer.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.vxref_sizeof.restype = u32
        L.vxref_generate_ray.argtypes = [u32, u32, u32, u32, vp]
        L.vxref_shade.restype = u32
        L.vxref_shade.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, u32, u32, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _ref = L
    return _ref


def _p(a):
    return a.ctypes.data if a is not None else None


# ---- software twin (tests/regression/raycast) through the reference's own object code ----
REF_RC_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libvxref_rc.so")
_ref_rc = None
RC_BUFFERS = ("tlas", "blas", "bvh", "tri", "triEx", "triIdx", "tex")


def have_ref_rc():
    return os.path.exists(REF_RC_SO)


def ref_rc():
    global _ref_rc
    if _ref_rc is None:
        L = C.CDLL(REF_RC_SO)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.rcref_scene_create.restype = vp
        L.rcref_scene_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int, C.c_int]
        L.rcref_scene_destroy.argtypes = [vp]
        L.rcref_scene_buffer.restype = u64
        L.rcref_scene_buffer.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.rcref_tlas_root.restype = u32
        L.rcref_tlas_root.argtypes = [vp]
        L.rcref_sizeof.restype = u32
        L.rcref_camera.argtypes = [vp, C.c_float, C.c_float, u32, u32, vp]
        L.rcref_render.restype = C.c_int
        L.rcref_render.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp]
        L.rcref_trace.argtypes = [vp, vp, vp, vp, vp, vp]
        _ref_rc = L
    return _ref_rc


def ref_rc_render_buffers(scene, w, h, y0, y1, cam14, light12=None, x_step=1, spp=1, max_depth=1):
    """The reference's own raycast render loop (render.h GenerateRay + Trace, compiled where it lies) over a raycast-format
    scene given as a dict of uint8 arrays (tlas, blas, bvh, tri, triEx, triIdx, tex, tlas_root) -- rows [y0,y1), every
    x_step-th column.  Returns (pixels [h, w] u32, primary rays traced).  BASELINE.md s2 baseline B1."""
    L = ref_rc()
    L.rcref_render_buffers.restype = C.c_uint64
    L.rcref_render_buffers.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    keep = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tri", "triEx", "triIdx", "bvh", "tlas", "blas", "tex")}
    ptrs = (C.c_void_p * 7)(*[keep[k].ctypes.data if keep[k].size else None for k in ("tri", "triEx", "triIdx", "bvh", "tlas", "blas", "tex")])
    cam = np.ascontiguousarray(cam14, np.float32)
    lig = np.ascontiguousarray(light12 if light12 is not None else RC_DEFAULT_LIGHT, np.float32)
    out = np.zeros((h, w), np.uint32)
    n = L.rcref_render_buffers(ptrs, int(scene["tlas_root"]), w, h, y0, y1, x_step, spp, max_depth, _p(cam), _p(lig), _p(out))
    return out, int(n)


class RcArgs(C.Structure):   # rc_args_t (oracle/rc_oracle.h)
    _fields_ = [("dst_width", C.c_uint32), ("dst_height", C.c_uint32),
                ("tri", C.c_void_p), ("triEx", C.c_void_p), ("triIdx", C.c_void_p), ("tex", C.c_void_p),
                ("bvh", C.c_void_p), ("blas", C.c_void_p), ("tlas", C.c_void_p), ("tlas_root", C.c_uint32),
                ("camera_pos", C.c_float * 3), ("camera_forward", C.c_float * 3), ("camera_right", C.c_float * 3),
                ("camera_up", C.c_float * 3), ("viewplane", C.c_float * 2),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("light_pos", C.c_float * 3), ("light_color", C.c_float * 3), ("ambient_color", C.c_float * 3),
                ("background_color", C.c_float * 3)]


RC_DEFAULT_LIGHT = (0.0, 10.0, -10.0, 1.0, 1.0, 1.0, 0.4, 0.4, 0.4, 0.4, 0.35, 0.25)   # raycast/main.cpp:29-32


def rc_args(scene, w, h, cam14, light12=RC_DEFAULT_LIGHT, spp=1, max_depth=1, tlas_root=None):
    """rc_args_t over the buffers of a raycast scene (dict of uint8 arrays: tlas, blas, bvh, tri, triEx, triIdx, tex).
    The returned object keeps the arrays alive."""
    a = RcArgs()
    keep = {k: np.ascontiguousarray(scene[k], np.uint8) for k in RC_BUFFERS}
    a.dst_width, a.dst_height = w, h
    for k in RC_BUFFERS:
        setattr(a, k, keep[k].ctypes.data if keep[k].size else None)
    a.tlas_root = int(scene["tlas_root"]) if tlas_root is None else tlas_root
    cam14 = [float(v) for v in cam14]
    a.camera_pos[:] = cam14[0:3]; a.camera_forward[:] = cam14[3:6]; a.camera_right[:] = cam14[6:9]
    a.camera_up[:] = cam14[9:12]; a.viewplane[:] = cam14[12:14]
    a.samples_per_pixel, a.max_depth = spp, max_depth
    l = [float(v) for v in light12]
    a.light_pos[:] = l[0:3]; a.light_color[:] = l[3:6]; a.ambient_color[:] = l[6:9]; a.background_color[:] = l[9:12]
    a._keep = keep
    return a


def rc_render(args, y0=0, y1=None):
    """oracle/rc_oracle.c:rc_render -> pixels [h, w] u32, colours [h, w, 3] f32"""
    L = orc()
    h, w = args.dst_height, args.dst_width
    y1 = h if y1 is None else y1
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h, w, 3), np.float32)
    L.rc_render.restype = C.c_int
    L.rc_render.argtypes = [C.POINTER(RcArgs), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    if L.rc_render(C.byref(args), y0, y1, _p(px), _p(col)) != 0:
        raise RuntimeError("rc_render: traversal stack would exceed BVH_STACK_SIZE")
    return px, col


def rc_render_mt(args, threads=None, rows_per_call=8):
    """rc_render over the whole frame from a thread pool: the C function writes only the rows it is given and ctypes releases the GIL, so
    disjoint row ranges share the output arrays (a 1080p frame of the 1M-triangle BVH2 in about a second on 16 cores)."""
    import concurrent.futures as cf
    L = orc()
    h, w = args.dst_height, args.dst_width
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h, w, 3), np.float32)
    L.rc_render.restype = C.c_int
    L.rc_render.argtypes = [C.POINTER(RcArgs), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]

    def one(r):
        return L.rc_render(C.byref(args), r[0], r[1], _p(px), _p(col))

    with cf.ThreadPoolExecutor(_threads(threads)) as ex:
        rc = list(ex.map(one, [(y, min(h, y + rows_per_call)) for y in range(0, h, rows_per_call)]))
    if any(rc):
        raise RuntimeError("rc_render: traversal stack would exceed BVH_STACK_SIZE")
    return px, col


def rc_trace(args, rays):
    L = orc()
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    out = np.zeros(len(rays), HIT_DTYPE)
    L.rc_trace.restype = C.c_int
    L.rc_trace.argtypes = [C.POINTER(RcArgs), C.c_void_p, C.c_void_p]
    for i in range(len(rays)):
        if L.rc_trace(C.byref(args), rays[i].ctypes.data, out[i:i + 1].ctypes.data) != 0:
            raise RuntimeError("rc_trace: stack")
    return out


def rc_camera_rays(args):
    L = orc()
    h, w = args.dst_height, args.dst_width
    out = np.zeros((h * w, 6), np.float32)
    L.rc_generate_ray.argtypes = [C.POINTER(RcArgs), C.c_uint32, C.c_uint32, C.c_void_p]
    for y in range(h):
        for x in range(w):
            L.rc_generate_ray(C.byref(args), x, y, out[y * w + x].ctypes.data)
    return out


class RefRcScene:
    """A scene built by the reference's raycast Scene/BVH/TLAS code (oracle/_ref/libvxref_rc.so)."""

    def __init__(self, objs, texs, refl, rotate=True):
        L = ref_rc()
        n = len(objs)
        a = (C.c_char_p * n)(*[o.encode() for o in objs])
        t = (C.c_char_p * n)(*[x.encode() for x in texs])
        r = (C.c_float * n)(*refl)
        self.h = L.rcref_scene_create(a, t, r, n, 1 if rotate else 0)
        if not self.h:
            raise RuntimeError("reference raycast scene build failed")
        self.buffers = {}
        for i, k in enumerate(RC_BUFFERS):
            ptr = C.c_void_p()
            size = L.rcref_scene_buffer(self.h, i, C.byref(ptr))
            self.buffers[k] = np.frombuffer((C.c_uint8 * size).from_address(ptr.value), np.uint8).copy() if size else np.zeros(0, np.uint8)
        self.tlas_root = int(L.rcref_tlas_root(self.h))

    def camera(self, vfov_deg, zoom, w, h):
        out = np.zeros(14, np.float32)
        ref_rc().rcref_camera(self.h, vfov_deg, zoom, w, h, _p(out))
        return out

    def render(self, w, h, spp, max_depth, cam14, light12):
        out = np.zeros((h, w), np.uint32)
        cam14 = np.ascontiguousarray(cam14, np.float32)
        light12 = np.ascontiguousarray(light12, np.float32)
        ref_rc().rcref_render(self.h, w, h, spp, max_depth, _p(cam14), _p(light12), _p(out))
        return out

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), HIT_DTYPE)
        d, bc, bi, ti = C.c_float(), (C.c_float * 3)(), C.c_uint32(), C.c_uint32()
        L = ref_rc()
        for i, r in enumerate(rays):
            rr = np.ascontiguousarray(r)
            L.rcref_trace(self.h, _p(rr), C.byref(d), bc, C.byref(bi), C.byref(ti))
            out[i] = (d.value, bc[0], bc[1], bc[2], bi.value, ti.value)
        return out

    def radiance(self, rays, max_depth, light12):
        """The reference's Trace (render.h:210-277, iterative mirror bounce) on arbitrary rays: (colours [n,3] f32, rgb8 [n] u32)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        light12 = np.ascontiguousarray(light12, np.float32)
        col = np.zeros((len(rays), 3), np.float32)
        px = np.zeros(len(rays), np.uint32)
        L = ref_rc()
        L.rcref_radiance.restype = C.c_int
        L.rcref_radiance.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rcref_radiance(self.h, _p(rays), len(rays), int(max_depth), _p(light12), _p(col), _p(px))
        return col, px

    def close(self):
        if self.h:
            ref_rc().rcref_scene_destroy(self.h)
            self.h = None


class Image:
    """Flat device-memory image + the four 32-bit DCR offsets (tracer.cpp:252-256), 64-byte aligned."""

    def __init__(self, scene):
        off = 64
        self.off = {}
        parts = []
        for k in ("tlas", "blas", "bvh", "tri"):
            b = np.ascontiguousarray(scene[k], np.uint8)
            self.off[k] = off
            parts.append((off, b))
            off = (off + b.size + 63) & ~63
        self.mem = np.zeros(off + 64, np.uint8)
        for o, b in parts:
            self.mem[o:o + b.size] = b


def _tmax(tmax):
    """Per-ray bounds as the C ABI defines them (include/vortex_hip.h, vxrt_trace): a bound above 1e30 is 1e30 -- the reference
    reports a missed box as 1e30 (rt_traversal.cpp:338) and relies on `d < hit.dist` (<= 1e30) to drop it."""
    if tmax is None:
        return None
    t = np.ascontiguousarray(tmax, np.float32).copy()
    t[t > np.float32(1e30)] = np.float32(1e30)
    return t


def trace_faithful(scene, rays, tmax=None, any_hit=False):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    img = scene if isinstance(scene, Image) else Image(scene)
    out = np.zeros(len(rays), HIT_DTYPE)
    st = OrcStats()
    tm = _tmax(tmax)
    orc().orc_trace_faithful(_p(img.mem), img.mem.size, img.off["tlas"], img.off["blas"], img.off["bvh"], img.off["tri"],
                             _p(rays), len(rays), _p(tm), _p(out), C.byref(st), int(any_hit))
    return out, st.as_dict()


def rng(seed, n, lib=None):
    """(hash, ints, floats) of orc_rng -- or of the reference's own helpers with lib=ref() (vxref_rng, common.h:129-147)."""
    L = lib or orc()
    fn = L.vxref_rng if lib is not None else L.orc_rng
    fn.restype = None
    fn.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    h, i, f = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
    fn(int(seed) & 0xFFFFFFFF, n, _p(h), _p(i), _p(f))
    return h, i, f


def stale_base_mask(scene, rays):
    """Per-ray flag: the reference expanded a TLAS internal node with a stale (BLAS) base_ptr for
    this ray (rt_traversal.cpp:91-92 after :119; see DESIGN.md 'reference quirks').  Such rays read
    unrelated memory in the reference and are outside the parity domain."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    img = scene if isinstance(scene, Image) else Image(scene)
    mask = np.zeros(len(rays), bool)
    for i in range(len(rays)):
        _, st = trace_faithful(img, rays[i:i + 1])
        mask[i] = st["stale_base"] != 0
    return mask


def trace_canonical(scene, rays, tmax=None, any_hit=False):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    out = np.zeros(len(rays), HIT_DTYPE)
    st = OrcStats()
    tm = _tmax(tmax)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri")}
    orc().orc_trace_canonical(_p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(rays), len(rays), _p(tm), _p(out),
                              C.byref(st), int(any_hit))
    return out, st.as_dict()


def trace_ref(scene, rays, any_hit=False):
    """The reference's own BVHTraverser (oracle/_ref)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    img = scene if isinstance(scene, Image) else Image(scene)
    out = np.zeros(len(rays), HIT_DTYPE)
    st = RefStats()
    ref().vxref_trace(_p(img.mem), img.mem.size, img.off["tlas"], img.off["blas"], img.off["bvh"], img.off["tri"],
                      _p(rays), len(rays), _p(out), C.byref(st), int(any_hit))
    return out, st.as_dict()


def camera_rays(w, h, y0=0, y1=None):
    y1 = h if y1 is None else y1
    out = np.zeros(((y1 - y0) * w, 6), np.float32)
    L = orc()
    L.orc_camera_rays.restype = None
    L.orc_camera_rays.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
    L.orc_camera_rays(w, h, y0, y1, _p(out))
    return out


def render(scene, w, h, params=None, y0=0, y1=None):
    """Whole RTU-test frame on the CPU restatement: pixels (h,w) u32, hits, colours."""
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    hits = np.zeros(h * w, HIT_DTYPE)
    col = np.zeros((h * w, 3), np.float32)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    orc().orc_render(w, h, y0, y1, _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                     _p(b["tex"]), C.byref(params), _p(px), _p(hits), _p(col))
    return px, hits.reshape(h, w), col.reshape(h, w, 3)


def render_ex(scene, w, h, params=None, shadow=0, y0=0, y1=None):
    """orc_render_ex: frame with the mirror bounce of closest.cpp:95-121 followed up to params.max_depth and,
    optionally, the shadow extension at every shaded hit.  Returns pixels, primary hits, colours, rays traced."""
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    hits = np.zeros(h * w, HIT_DTYPE)
    col = np.zeros((h * w, 3), np.float32)
    n = C.c_uint64(0)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    orc().orc_render_ex(w, h, y0, y1, _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                        _p(b["tex"]), C.byref(params), shadow, _p(px), _p(hits), _p(col), C.byref(n))
    return px, hits.reshape(h, w), col.reshape(h, w, 3), int(n.value)


def render_ao(scene, w, h, params=None, spp=4, radius=10.0, seed=0, y0=0, y1=None):
    """orc_render_ao: primary hit -> Lambert colour x fraction of `spp` cosine-weighted occlusion rays
    (tmax = radius) that reach nothing.  Returns pixels, colours, unoccluded counts, rays traced."""
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h * w, 3), np.float32)
    cnt = np.zeros((h, w), np.uint32)
    n = C.c_uint64(0)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    orc().orc_render_ao(w, h, y0, y1, _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                        _p(b["tex"]), C.byref(params), spp, radius, seed, _p(px), _p(col), _p(cnt), C.byref(n))
    return px, col.reshape(h, w, 3), cnt, int(n.value)


def render_gi(scene, w, h, params=None, seed=0, y0=0, y1=None):
    """orc_render_gi: primary hit + one cosine-weighted diffuse bounce.  Returns pixels, colours, rays traced."""
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h * w, 3), np.float32)
    n = C.c_uint64(0)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    L = orc()
    L.orc_render_gi.restype = C.c_int
    L.orc_render_gi.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 7 + [C.POINTER(ShadeParams), C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.orc_render_gi(w, h, y0, y1, _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                    _p(b["tex"]), C.byref(params), seed, _p(px), _p(col), C.byref(n))
    return px, col.reshape(h, w, 3), int(n.value)


def _threads(threads=None):
    if threads:
        return max(1, int(threads))
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def _row_chunks(y0, y1, ranges=None, rows_per_call=4):
    """Row ranges for the threaded checkers: small chunks dealt to a pool, so that cheap (sky) and expensive rows even out.
    ranges: optional list of (y0, y1) row ranges to render instead of the single range [y0, y1)."""
    step = max(1, rows_per_call)
    out = []
    for a, b in (ranges if ranges is not None else [(y0, y1)]):
        out += [(y, min(y + step, b)) for y in range(a, b, step)]
    return out


def trace_mt(fn, scene, rays, threads=None, tmax=None, **kw):
    """fn = trace_faithful / trace_canonical / trace_ref over a ray buffer from a thread pool (the foreign calls release the GIL; every
    call owns its output).  Returns the hit records only."""
    import concurrent.futures as cf
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    if fn is not trace_canonical and not isinstance(scene, Image):
        scene = Image(scene)
    n = len(rays)
    t = _threads(threads)
    cuts = np.linspace(0, n, min(n, t * 8) + 1).astype(np.int64) if n else np.zeros(1, np.int64)

    def one(i):
        a, b = int(cuts[i]), int(cuts[i + 1])
        if tmax is not None and fn is not trace_ref:
            return fn(scene, rays[a:b], tmax=tmax[a:b], **kw)[0]
        return fn(scene, rays[a:b], **kw)[0]

    with cf.ThreadPoolExecutor(t) as ex:
        parts = list(ex.map(one, range(len(cuts) - 1)))
    return np.concatenate(parts) if parts else np.zeros(0, HIT_DTYPE)


def render_ex_mt(scene, w, h, params=None, shadow=0, y0=0, y1=None, threads=None, ranges=None):
    """render_ex over rows [y0, y1) from a thread pool: orc_render_ex writes only the rows it is given and ctypes releases the GIL
    for the foreign call, so disjoint row ranges share the output arrays.  Makes WHOLE full-size frames checkable in seconds
    (tests/test_gpu_configs.py); same arithmetic as render_ex -- one call per row range instead of one call."""
    import concurrent.futures as cf
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    hits = np.zeros(h * w, HIT_DTYPE)
    col = np.zeros((h * w, 3), np.float32)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    L = orc()

    def one(r):
        n = C.c_uint64(0)
        L.orc_render_ex(w, h, r[0], r[1], _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                        _p(b["tex"]), C.byref(params), shadow, _p(px), _p(hits), _p(col), C.byref(n))
        return int(n.value)

    with cf.ThreadPoolExecutor(_threads(threads)) as ex:
        total = sum(ex.map(one, _row_chunks(y0, y1, ranges)))
    return px, hits.reshape(h, w), col.reshape(h, w, 3), total


def render_gi_mt(scene, w, h, params=None, seed=0, y0=0, y1=None, threads=None, ranges=None):
    """render_gi over rows [y0, y1) from a thread pool (see render_ex_mt)."""
    import concurrent.futures as cf
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h * w, 3), np.float32)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    L = orc()
    L.orc_render_gi.restype = C.c_int
    L.orc_render_gi.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 7 + [C.POINTER(ShadeParams), C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]

    def one(r):
        n = C.c_uint64(0)
        L.orc_render_gi(w, h, r[0], r[1], _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                        _p(b["tex"]), C.byref(params), seed, _p(px), _p(col), C.byref(n))
        return int(n.value)

    with cf.ThreadPoolExecutor(_threads(threads)) as ex:
        total = sum(ex.map(one, _row_chunks(y0, y1, ranges)))
    return px, col.reshape(h, w, 3), total


def render_ao_mt(scene, w, h, params=None, spp=4, radius=10.0, seed=0, y0=0, y1=None, threads=None, ranges=None):
    """render_ao over rows [y0, y1) from a thread pool (see render_ex_mt)."""
    import concurrent.futures as cf
    y1 = h if y1 is None else y1
    params = params or shade_params()
    px = np.zeros((h, w), np.uint32)
    col = np.zeros((h * w, 3), np.float32)
    cnt = np.zeros((h, w), np.uint32)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
    L = orc()

    def one(r):
        n = C.c_uint64(0)
        L.orc_render_ao(w, h, r[0], r[1], _p(b["tlas"]), _p(b["blas"]), _p(b["bvh"]), _p(b["tri"]), _p(b["triEx"]), _p(b["mat"]),
                        _p(b["tex"]), C.byref(params), spp, radius, seed, _p(px), _p(col), _p(cnt), C.byref(n))
        return int(n.value)

    with cf.ThreadPoolExecutor(_threads(threads)) as ex:
        total = sum(ex.map(one, _row_chunks(y0, y1, ranges, 1)))
    return px, col.reshape(h, w, 3), cnt, total


def shade(scene, rays, hits, params=None):
    params = params or shade_params()
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    hits = np.ascontiguousarray(hits)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("blas", "triEx", "mat", "tex")}
    col = np.zeros((len(rays), 3), np.float32)
    L = orc()
    for i in range(len(rays)):
        L.orc_shade(rays[i].ctypes.data, hits[i:i + 1].ctypes.data, _p(b["blas"]), _p(b["triEx"]), _p(b["mat"]), _p(b["tex"]),
                    C.byref(params), col[i].ctypes.data)
    px = np.array([L.orc_pack_rgb8(col[i].ctypes.data) for i in range(len(rays))], np.uint32)
    return col, px


def ref_scene(obj_paths):
    """Run the reference's own scene builder on OBJ files; returns dict of byte buffers."""
    L = ref()
    arr = (C.c_char_p * len(obj_paths))(*[str(p).encode() for p in obj_paths])
    h = L.vxref_scene_create(arr, len(obj_paths))
    if not h:
        raise RuntimeError("reference scene builder failed")
    try:
        out = {}
        for i, k in enumerate(("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")):
            p = C.c_void_p()
            n = L.vxref_scene_buffer(h, i, C.byref(p))
            out[k] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        return out
    finally:
        L.vxref_scene_destroy(h)


def ref_shade(scene, rays, hits, params=None):
    params = params or shade_params()
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    b = {k: np.ascontiguousarray(scene[k], np.uint8) for k in ("blas", "triEx", "mat", "tex")}
    L = ref()
    col = np.zeros((len(rays), 3), np.float32)
    px = np.zeros(len(rays), np.uint32)
    for i in range(len(rays)):
        hh = hits[i]
        px[i] = L.vxref_shade(rays[i].ctypes.data, float(hh["dist"]), float(hh["bx"]), float(hh["by"]), float(hh["bz"]),
                              int(hh["blasIdx"]), int(hh["triIdx"]), int(hh["dist"] != np.float32(1e30)),
                              _p(b["blas"]), _p(b["triEx"]), _p(b["mat"]), _p(b["tex"]),
                              C.cast(params.ambient, C.c_void_p), C.cast(params.light_color, C.c_void_p),
                              C.cast(params.light_pos, C.c_void_p), C.cast(params.background, C.c_void_p), col[i].ctypes.data)
    return col, px
