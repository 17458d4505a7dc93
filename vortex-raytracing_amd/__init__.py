"""vortex-raytracing_amd -- MI355X-native hot path of the Vortex ray-tracing test.

Python here is only a thin ctypes host over the C ABI in include/vortex_hip.h:
  runtime  -- the vx_* host API (reference runtime/include/vortex.h:80-145) through libvortex.so,
              which loads the backend libvortex-hip.so exactly as the reference dispatcher does
  rtapi    -- the vxrt_* direct launch API on caller-owned device memory / streams
  scene    -- BVH4 builder, quantiser and procedural scenes (libvxrt_scene.so)
  tracer   -- host-side mirror of the reference Tracer (tests/regression/raytracing/tracer.cpp)

There is no CPU fallback: importing works without a GPU (for build checks), but every compute
entry raises if the HIP library is missing or no device is present.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("VXRT_LIB_DIR") or os.path.join(PKG_DIR, "lib")   # (override: A/B of prebuilt kernel variants, tools/build_variants.py)
VXBIN_DIR = os.path.join(PKG_DIR, "vxbin")
REPO_DIR = os.path.dirname(PKG_DIR)


def lib_path(name):
    p = os.path.join(LIB_DIR, name)
    if not os.path.exists(p):
        raise RuntimeError(
            "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)" % p)
    return p


from . import scene, runtime, rtapi, tracer, sharding  # noqa: E402,F401
