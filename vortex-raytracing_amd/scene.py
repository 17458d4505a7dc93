"""ctypes binding of libvxrt_scene.so (csrc/scene_builder.cpp): scene construction on the host.

Produces the buffers Tracer::init/setup upload (reference tracer.cpp:124-161,217-241) in the
reference's byte formats."""
import ctypes as C
import os

import numpy as np

from . import lib_path

NODE_BYTES, BLAS_BYTES, TRI_BYTES, TRIEX_BYTES, MAT_BYTES = 52, 160, 36, 64, 88
_BUFFERS = ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex", "triIdx")
_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(lib_path("libvxrt_scene.so"))
        lib.vxs_scene_create_procedural.restype = C.c_void_p
        lib.vxs_scene_create_procedural.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.vxs_scene_create_from_tris.restype = C.c_void_p
        lib.vxs_scene_create_from_tris.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.vxs_scene_load_obj.restype = C.c_void_p
        lib.vxs_scene_load_obj.argtypes = [C.c_char_p, C.c_uint32]
        lib.vxs_scene_destroy.argtypes = [C.c_void_p]
        lib.vxs_scene_buffer.restype = C.c_uint64
        lib.vxs_scene_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.vxs_scene_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


class Scene:
    """Host copy of the scene buffers as numpy uint8 arrays (owned by Python after construction)."""

    def __init__(self, buffers, info=None, bounds=None, name=""):
        self.buffers = {k: np.ascontiguousarray(v, dtype=np.uint8) for k, v in buffers.items()}
        self.info = info or {}
        self.bounds = bounds
        self.name = name

    def __getitem__(self, k):
        return self.buffers[k]

    @property
    def n_tris(self):
        return self.buffers["tri"].size // TRI_BYTES

    @property
    def n_bvh_nodes(self):
        return self.buffers["bvh"].size // NODE_BYTES

    @property
    def n_tlas_nodes(self):
        return self.buffers["tlas"].size // NODE_BYTES

    @property
    def n_blas(self):
        return self.buffers["blas"].size // BLAS_BYTES

    @property
    def n_mats(self):
        return self.buffers["mat"].size // MAT_BYTES

    def save(self, path):
        np.savez_compressed(path, **self.buffers)

    @staticmethod
    def load(path, name=""):
        with np.load(path) as z:
            return Scene({k: z[k] for k in z.files}, name=name)


def _from_handle(h, name):
    lib = _load()
    if not h:
        raise RuntimeError("scene construction failed: %s" % name)
    try:
        bufs = {}
        for i, k in enumerate(_BUFFERS):
            p = C.c_void_p()
            n = lib.vxs_scene_buffer(h, i, C.byref(p))
            bufs[k] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        info = (C.c_uint32 * 6)()
        bounds = (C.c_float * 6)()
        lib.vxs_scene_info(h, info, bounds)
        keys = ("max_depth", "n_leaves", "max_leaf", "n_bvh_nodes", "n_tlas_nodes", "n_tris")
        return Scene(bufs, dict(zip(keys, list(info))), np.array(list(bounds), np.float32), name)
    finally:
        lib.vxs_scene_destroy(h)


def _cache_path(name, a, b, seed):
    """File of a built scene under $VXRT_SCENE_CACHE (unset: no cache).  The key holds the builder's knobs (VXS_*) and the library's
    build time: measurement scripts that start a dozen processes over the same 1M-triangle scene build it once."""
    d = os.environ.get("VXRT_SCENE_CACHE")
    if not d:
        return None
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvxrt_scene.so")
    key = repr((name, a, b, seed, sorted((k, v) for k, v in os.environ.items() if k.startswith("VXS_")), os.path.getmtime(lib)))
    return os.path.join(d, "scene_%s_%s.npz" % (name, hashlib.sha1(key.encode()).hexdigest()[:16]))


def procedural(name, a=0, b=0, seed=1):
    """'cornell' | 'blob' (a = icosphere subdivisions) | 'atrium' (a = level; 8 -> 1,048,576 tris)
    | 'hairball' (a strands x b segments)."""
    path = _cache_path(name, a, b, seed)
    if path and os.path.exists(path):
        with np.load(path) as z:
            info = dict(zip(("max_depth", "n_leaves", "max_leaf", "n_bvh_nodes", "n_tlas_nodes", "n_tris"), z["__info"].tolist()))
            return Scene({k: z[k] for k in z.files if not k.startswith("__")}, info, z["__bounds"].copy(), name)
    sc = _from_handle(_load().vxs_scene_create_procedural(name.encode(), a, b, seed), name)
    if path:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, __info=np.array([sc.info[k] for k in ("max_depth", "n_leaves", "max_leaf", "n_bvh_nodes", "n_tlas_nodes", "n_tris")], np.int64),
                 __bounds=np.asarray(sc.bounds, np.float32), **sc.buffers)
        os.replace(tmp, path)
    return sc


def from_triangles(meshes, transforms=None):
    """meshes: list of float32 arrays [n_i, 9] (v0,v1,v2).  transforms: optional list of 4x4."""
    ntris = np.array([len(m) for m in meshes], np.uint32)
    tris = np.ascontiguousarray(np.concatenate([np.asarray(m, np.float32).reshape(-1, 9) for m in meshes]), np.float32)
    tr = None
    if transforms is not None:
        tr = np.ascontiguousarray(np.stack([np.asarray(t, np.float32).reshape(16) for t in transforms]), np.float32)
    h = _load().vxs_scene_create_from_tris(len(meshes), ntris.ctypes.data, tris.ctypes.data, None,
                                           tr.ctypes.data if tr is not None else None, None, None)
    return _from_handle(h, "triangles")


def image_load(path):
    """Decode an image file as the scene ingest does (PNG, PPM/PGM): uint32 array [h, w] of 0x00RRGGBB texels."""
    L = _load()
    w, h = C.c_uint32(0), C.c_uint32(0)
    L.vxs_image_load.restype = C.c_int
    L.vxs_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]
    if L.vxs_image_load(str(path).encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError("cannot decode image %s" % path)
    out = np.zeros((h.value, w.value), np.uint32)
    if L.vxs_image_load(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data, out.size) != 0:
        raise ValueError("cannot decode image %s" % path)
    return out


def load_obj(path, instances=1):
    return _from_handle(_load().vxs_scene_load_obj(str(path).encode(), instances), str(path))


RC_BUFFERS = ("tlas", "blas", "bvh", "tri", "triEx", "triIdx", "tex")


def rc_procedural(name, a=0, b=0, seed=1, copies=1, reflectivity=None):
    """Procedural scene in the formats of the software twin (tests/regression/raycast/common.h): dict of uint8 arrays
    tlas / blas / bvh / tri / triEx / triIdx / tex plus 'tlas_root', 'bounds', 'max_depth'."""
    L = _load()
    L.vxs_rc_scene_create_procedural.restype = C.c_void_p
    L.vxs_rc_scene_create_procedural.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.vxs_rc_scene_destroy.argtypes = [C.c_void_p]
    L.vxs_rc_scene_buffer.restype = C.c_uint64
    L.vxs_rc_scene_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.vxs_rc_scene_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    refl = None
    if reflectivity is not None:
        refl = np.ascontiguousarray(list(reflectivity) + [0.0] * max(0, copies - len(reflectivity)), np.float32)
    h = L.vxs_rc_scene_create_procedural(name.encode(), a, b, seed, copies, refl.ctypes.data if refl is not None else None)
    if not h:
        raise RuntimeError("raycast-format scene construction failed: %s" % name)
    try:
        out = {}
        for i, k in enumerate(RC_BUFFERS):
            p = C.c_void_p()
            n = L.vxs_rc_scene_buffer(h, i, C.byref(p))
            out[k] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        info = (C.c_uint32 * 2)()
        bounds = (C.c_float * 6)()
        L.vxs_rc_scene_info(h, info, bounds)
        out["tlas_root"] = int(info[0])
        out["max_depth"] = int(info[1])
        out["bounds"] = np.array(list(bounds), np.float32)
        return out
    finally:
        L.vxs_rc_scene_destroy(h)


def rc_camera_like_rtu(width, height):
    """cam14 (pos, forward, right, up, viewplane) that frames what the RTU kernel's fixed camera sees
    (raytracing/kernel.cpp:28-39: eye (0,100,0), looking along +x, u in [-W/H, W/H], v in [-1, 1])."""
    return np.array([0, 100, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 2.0 * width / height, 2.0], np.float32)
