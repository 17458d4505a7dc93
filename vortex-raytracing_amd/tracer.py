"""Host-side mirror of the reference's Tracer (tests/regression/raytracing/tracer.{h,cpp}) written
against the vx_* API only, so it exercises exactly the calls the reference host program makes:
init (:90-169) = open device, upload the 4 kernel images, 11 vx_mem_alloc + vx_mem_address;
setup (:171-260) = 9 vx_copy_to_dev, shader binding table, 4 DCR writes; run (:262-288) =
vx_upload_bytes(kernel_arg), vx_start, vx_ready_wait, vx_copy_from_dev.

DeviceScene is the equivalent for the vxrt_* direct API: scene buffers as torch uint8 CUDA tensors."""
import ctypes as C
import os
import struct

import numpy as np

from . import VXBIN_DIR, runtime, rtapi
from .scene import NODE_BYTES, BLAS_BYTES, TRI_BYTES, MAT_BYTES

KERNEL_ARG_FMT = "<IIQ" + "Q" * 9 + "I" + "3f3f3f3f" + "2f" + "II" + "3f3f3f3f" + "4xQ"
assert struct.calcsize(KERNEL_ARG_FMT) == 216  # raytracing/common.h:164-195

# main.cpp:34-41
DEFAULT_LIGHT_POS = (0.0, 10.0, -10.0)
DEFAULT_LIGHT_COLOR = (1.0, 1.0, 1.0)
DEFAULT_AMBIENT = (0.4, 0.4, 0.4)
DEFAULT_BACKGROUND = (0.4, 0.35, 0.25)


class Tracer:
    def __init__(self, width, height, samples_per_pixel=1, max_depth=1):
        self.width, self.height = width, height
        self.spp, self.max_depth = samples_per_pixel, max_depth
        self.dev = None
        self.bufs = {}
        self.args = None

    def init(self, scene, vxbin_dir=VXBIN_DIR):
        self.scene = scene
        d = self.dev = runtime.Device()
        self.krnl = d.upload_kernel_file(os.path.join(vxbin_dir, "kernel.vxbin"))
        self.miss = d.upload_kernel_file(os.path.join(vxbin_dir, "miss.vxbin"))
        self.closest = d.upload_kernel_file(os.path.join(vxbin_dir, "closest.vxbin"))
        self.anyhit = d.upload_kernel_file(os.path.join(vxbin_dir, "anyhit.vxbin"))
        for k in ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "mat", "tex"):
            self.bufs[k] = d.mem_alloc(int(scene[k].size), runtime.VX_MEM_READ)
        # the reference also uploads the un-quantised BVH2-style nodes (tracer.cpp:144-145); the RTU
        # never reads them, so the mirror allocates a token buffer to keep kernel_arg_t complete
        self.bufs["bvh2"] = d.mem_alloc(64, runtime.VX_MEM_READ)
        self.bufs["out"] = d.mem_alloc(self.width * self.height * 4, runtime.VX_MEM_WRITE)
        self.bufs["sbt"] = d.mem_alloc(32, runtime.VX_MEM_READ)
        return 0

    def setup(self, light_pos=DEFAULT_LIGHT_POS, light_color=DEFAULT_LIGHT_COLOR, ambient=DEFAULT_AMBIENT,
              background=DEFAULT_BACKGROUND, row_window=None, shadow=False):
        d = self.dev
        for k in ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "mat", "tex"):
            self.bufs[k].write(self.scene[k])
        sbt = struct.pack("<4Q", self.miss.address, self.closest.address, 0, self.anyhit.address)
        self.bufs["sbt"].write(sbt)
        a = {k: b.address for k, b in self.bufs.items()}
        d.dcr_write(runtime.VX_DCR_BASE_RTX_TLAS_PTR, a["tlas"])   # tracer.cpp:252-256 (truncated to 32 bit)
        d.dcr_write(runtime.VX_DCR_BASE_RTX_BLAS_PTR, a["blas"])
        d.dcr_write(runtime.VX_DCR_BASE_RTX_BVH_PTR, a["bvh"])
        d.dcr_write(runtime.VX_DCR_BASE_RTX_TRI_PTR, a["tri"])
        y0, y1 = row_window if row_window else (0, 0)
        d.dcr_write(runtime.VX_DCR_HIP_ROW_BEGIN, y0)
        d.dcr_write(runtime.VX_DCR_HIP_ROW_END, y1)
        d.dcr_write(runtime.VX_DCR_HIP_SHADOW_RAYS, 1 if shadow else 0)
        self.kernel_arg = struct.pack(
            KERNEL_ARG_FMT, self.width, self.height, a["out"], a["tri"], a["triEx"], a["triIdx"], a["mat"], a["tex"],
            a["bvh2"], a["bvh"], a["blas"], a["tlas"], 0,
            0.0, 100.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0,   # camera_* are unused by the RTU kernel
            1.0, 1.0, self.spp, self.max_depth, *light_pos, *light_color, *ambient, *background, a["sbt"])
        return 0

    def run(self):
        d = self.dev
        if self.args is not None:
            self.args.free()
        self.args = d.upload_bytes(self.kernel_arg)
        d.start(self.krnl, self.args)
        d.ready_wait(runtime.VX_MAX_TIMEOUT)
        raw = self.bufs["out"].read()
        return np.frombuffer(raw, dtype=np.uint32).reshape(self.height, self.width).copy()

    def close(self):
        if self.dev is None:
            return
        for b in list(self.bufs.values()) + [self.args, self.krnl, self.miss, self.closest, self.anyhit]:
            if b is not None:
                b.free()
        self.bufs = {}
        self.dev.close()
        self.dev = None


class DeviceScene:
    """Scene buffers resident in HBM as torch uint8 tensors + the vxrt_scene_t that points at them."""

    def __init__(self, scene, device="cuda:0"):
        import torch
        self.t = {k: torch.from_numpy(np.ascontiguousarray(scene[k])).to(device) for k in ("tlas", "blas", "bvh", "tri", "triEx", "mat", "tex")}
        s = rtapi.VxrtScene()
        for k, t in self.t.items():
            setattr(s, k, t.data_ptr() if t.numel() else None)
        s.n_tlas_nodes = scene["tlas"].size // NODE_BYTES
        s.n_blas = scene["blas"].size // BLAS_BYTES
        s.n_bvh_nodes = scene["bvh"].size // NODE_BYTES
        s.n_tris = scene["tri"].size // TRI_BYTES
        s.n_mats = scene["mat"].size // MAT_BYTES
        s.tex_bytes = scene["tex"].size
        self.c = s
        self.device = device
        # one-time re-layout on the GPU (decoded boxes, inlined leaves, edge-form triangles)
        with torch.cuda.device(device):
            self.accel = rtapi.accel_build(s, torch.cuda.current_stream().cuda_stream)

    @classmethod
    def build_on_gpu(cls, tri, triEx=None, mat=None, tex=None, device="cuda:0", leaf_max=0, transforms=None):
        """Scene whose BVHs are built on the GPU from triangle soups: every mesh's BLAS with vxrt_bvh_build, the TLAS over the
        instances with vxrt_tlas_build.  tri: float32 [n, 9] (tri_t) or a list of such arrays (one per mesh); triEx: uint8 / float32
        [n, 64 B] (tri_ex_t) or a list (default: flat normals, material 0); transforms: optional list of 4x4 object-to-world
        matrices (default identity); mat / tex as the reference's buffers (default: one grey material, no texture).  Instance
        records are what the reference's scene builder writes (scene.cpp:84-99).  The buffers never exist on the host; .bvh_info
        holds the last mesh's builder counts, .tlas_info the TLAS's."""
        import torch
        self = cls.__new__(cls)
        self.device = device
        meshes = [np.ascontiguousarray(np.asarray(t, np.float32).reshape(-1, 9)) for t in (tri if isinstance(tri, (list, tuple)) else [tri])]
        exs = list(triEx) if isinstance(triEx, (list, tuple)) else [triEx] * len(meshes) if triEx is None else [triEx]
        if transforms is None:
            transforms = [np.eye(4, dtype=np.float32)] * len(meshes)
        assert len(exs) == len(meshes) == len(transforms)
        for i, (m, e) in enumerate(zip(meshes, exs)):
            if e is None:
                v = m.reshape(-1, 3, 3)
                nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
                nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)).astype(np.float32)
                ex = np.zeros((len(m), 16), np.float32)
                ex[:, 0:3] = ex[:, 3:6] = ex[:, 6:9] = nrm
                exs[i] = ex
            exs[i] = np.ascontiguousarray(exs[i]).view(np.uint8).reshape(-1, 64)
            assert len(exs[i]) == len(m)
        n = sum(len(m) for m in meshes)
        with torch.cuda.device(device):
            t_tri = torch.from_numpy(np.concatenate(meshes)).to(device).view(torch.uint8).reshape(-1)
            t_ex = torch.from_numpy(np.concatenate(exs).reshape(-1)).to(device)
            if mat is None:
                m0 = np.zeros(22, np.float32)
                m0[3:6] = 0.8
                mat = m0.view(np.uint8).copy()
                mat[64:68] = np.frombuffer(struct.pack("<i", -1), np.uint8)
            t_mat = torch.from_numpy(np.ascontiguousarray(mat).view(np.uint8).reshape(-1)).to(device)
            t_tex = torch.from_numpy(np.ascontiguousarray(tex if tex is not None and np.asarray(tex).size else np.zeros(4, np.uint8)).view(np.uint8).reshape(-1)).to(device)
            cap = 2 * n
            t_bvh = torch.zeros(cap * NODE_BYTES, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream().cuda_stream
            blas = np.zeros((len(meshes), 40), np.float32)
            boxes = np.zeros((len(meshes), 6), np.float32)
            tri_off = node_off = 0
            for i, m in enumerate(meshes):
                info = rtapi.bvh_build(t_tri.data_ptr() + tri_off * TRI_BYTES, t_ex.data_ptr() + tri_off * 64, len(m),
                                       t_bvh.data_ptr() + node_off * NODE_BYTES, cap - node_off, tri_off, leaf_max, stream)
                xf = np.asarray(transforms[i], np.float64).reshape(4, 4)
                blas[i].view(np.uint32)[0] = node_off                               # bvh_offset
                blas[i, 1:17] = np.linalg.inv(xf).astype(np.float32).reshape(-1)    # invTransform
                blas[i, 17:33] = xf.astype(np.float32).reshape(-1)                  # transform
                b = np.array(list(info.bounds), np.float64)
                corners = np.array([[b[3 * ((c >> k) & 1) + k] for k in range(3)] + [1.0] for c in range(8)])
                wc = (corners @ xf.T)[:, :3].astype(np.float32)                     # world-space bounds of the instance (bvh.cpp:295-304)
                boxes[i, :3], boxes[i, 3:] = wc.min(0), wc.max(0)
                tri_off += len(m)
                node_off += info.n_nodes
            t_bvh = t_bvh[: node_off * NODE_BYTES]
            t_boxes = torch.from_numpy(boxes).to(device)
            t_tlas = torch.zeros(max(1, 2 * len(meshes)) * NODE_BYTES, dtype=torch.uint8, device=device)
            tinfo = rtapi.tlas_build(t_boxes.data_ptr(), len(meshes), t_tlas.data_ptr(), max(1, 2 * len(meshes)), stream)
            t_tlas = t_tlas[: tinfo.n_nodes * NODE_BYTES]
            t_blas = torch.from_numpy(blas.view(np.uint8).reshape(-1).copy()).to(device)
            self.t = {"tlas": t_tlas, "blas": t_blas, "bvh": t_bvh, "tri": t_tri, "triEx": t_ex, "mat": t_mat, "tex": t_tex}
            s = rtapi.VxrtScene()
            for k, t in self.t.items():
                setattr(s, k, t.data_ptr())
            s.n_tlas_nodes, s.n_blas, s.n_bvh_nodes, s.n_tris = tinfo.n_nodes, len(meshes), node_off, n
            s.n_mats = t_mat.numel() // MAT_BYTES
            s.tex_bytes = t_tex.numel()
            self.c = s
            self.bvh_info, self.tlas_info = info, tinfo
            self.accel = rtapi.accel_build(s, stream)
        return self

    def to_host(self):
        """The scene buffers as a scene.Scene (numpy), e.g. to hand a GPU-built tree to the oracle."""
        from .scene import Scene
        bufs = {k: t.cpu().numpy() for k, t in self.t.items()}
        bufs["triIdx"] = np.arange(bufs["tri"].size // TRI_BYTES, dtype=np.uint32).view(np.uint8)
        return Scene(bufs, name="device")

    def close(self):
        if getattr(self, "accel", None):
            rtapi.accel_destroy(self.accel)
            self.accel = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ppm(pixels, path):
    """ASCII P3 writer with the reference's byte order and vertical flip (tracer.cpp:15-33)."""
    h, w = pixels.shape
    px = pixels[::-1].reshape(-1)
    b0, b1, b2 = px & 255, (px >> 8) & 255, (px >> 16) & 255
    with open(path, "w") as f:
        f.write("P3\n%d %d\n255\n" % (w, h))
        # the reference streams float(byte) through operator<<, i.e. "b g r" of the little-endian pixel bytes 2,1,0
        np.savetxt(f, np.stack([b2, b1, b0], axis=1), fmt="%d")


# ---------------------------------------------------------------------------------------------
# software twin: tests/regression/raycast/tracer.cpp through the same vx_* calls
# ---------------------------------------------------------------------------------------------
RC_KERNEL_ARG_FMT = "<IIQ7QI12f2fII12fI"   # raycast/common.h:126-150, 192 bytes
assert struct.calcsize(RC_KERNEL_ARG_FMT) == 192
RC_VXBIN_DIR = os.path.join(VXBIN_DIR, "raycast")
RC_BUFFERS = ("tri", "triEx", "triIdx", "tlas", "blas", "bvh", "tex")


class RaycastTracer:
    """Call sequence of raycast/tracer.cpp (init :107-150, setup :168-224, run :226-247) on buffers already in the
    reference's formats (dict of uint8 arrays + tlas_root), with camera/light as Tracer::setup computes them."""

    def __init__(self, width, height, samples_per_pixel=1, max_depth=1):
        self.width, self.height, self.spp, self.max_depth = width, height, samples_per_pixel, max_depth
        self.dev, self.bufs, self.args = None, {}, None

    def init(self, scene, vxbin_dir=RC_VXBIN_DIR):
        self.scene = scene
        d = self.dev = runtime.Device()
        self.krnl = d.upload_kernel_file(os.path.join(vxbin_dir, "kernel.vxbin"))
        for k in RC_BUFFERS:
            self.bufs[k] = d.mem_alloc(int(scene[k].size), runtime.VX_MEM_READ)
        self.bufs["out"] = d.mem_alloc(self.width * self.height * 4, runtime.VX_MEM_WRITE)
        return 0

    def setup(self, cam14, light12, row_window=None):
        d = self.dev
        for k in RC_BUFFERS:
            self.bufs[k].write(self.scene[k])
        a = {k: b.address for k, b in self.bufs.items()}
        y0, y1 = row_window if row_window else (0, 0)
        d.dcr_write(runtime.VX_DCR_HIP_ROW_BEGIN, y0)
        d.dcr_write(runtime.VX_DCR_HIP_ROW_END, y1)
        self.kernel_arg = struct.pack(
            RC_KERNEL_ARG_FMT, self.width, self.height, a["out"], a["tri"], a["triEx"], a["triIdx"], a["tex"], a["bvh"], a["blas"], a["tlas"],
            int(self.scene["tlas_root"]), *[float(v) for v in cam14[:12]], float(cam14[12]), float(cam14[13]), self.spp, self.max_depth,
            *[float(v) for v in light12], 0)
        return 0

    def run(self):
        d = self.dev
        if self.args is not None:
            self.args.free()
        self.args = d.upload_bytes(self.kernel_arg)
        d.start(self.krnl, self.args)
        d.ready_wait(runtime.VX_MAX_TIMEOUT)
        raw = self.bufs["out"].read()
        return np.frombuffer(raw, dtype=np.uint32).reshape(self.height, self.width).copy()

    def close(self):
        if self.dev is None:
            return
        for b in list(self.bufs.values()) + [self.args, self.krnl]:
            if b is not None:
                b.free()
        self.bufs = {}
        self.dev.close()
        self.dev = None


class RcDeviceScene:
    """Raycast scene buffers resident in HBM as torch uint8 tensors + the vxrc_scene_t that points at them."""

    def __init__(self, scene, device="cuda:0"):
        import torch
        self.device = device
        self.t = {k: torch.from_numpy(np.ascontiguousarray(scene[k], np.uint8)).to(device) for k in RC_BUFFERS}
        s = rtapi.RcScene()
        for k, t in self.t.items():
            setattr(s, k, t.data_ptr() if t.numel() else None)
        s.n_tlas_nodes = scene["tlas"].size // 32
        s.n_blas = scene["blas"].size // 160
        s.n_bvh_nodes = scene["bvh"].size // 32
        s.n_tris = min(scene["tri"].size // 36, scene["triEx"].size // 60)
        s.n_tri_idx = scene["triIdx"].size // 4
        s.tlas_root = int(scene["tlas_root"])
        s.tex_bytes = scene["tex"].size
        self.c = s
        self._accel = None

    @property
    def accel(self):
        """The twin's acceleration layout (vxrc_accel_build), built on first use; raises VxError for a malformed scene."""
        if self._accel is None:
            import torch
            with torch.cuda.device(self.device):
                self._accel = rtapi.rc_accel_build(self.c, torch.cuda.current_stream().cuda_stream)
        return self._accel

    def close(self):
        if getattr(self, "_accel", None):
            rtapi.rc_accel_destroy(self._accel)
            self._accel = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
