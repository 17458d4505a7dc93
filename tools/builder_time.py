#!/usr/bin/env python3
"""Host-side build of the two big procedural scenes with the builder's own phase timings (VXS_VERBOSE) and a hash of the result.
usage: tools/builder_time.py [atrium|hairball|both]"""
import hashlib, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("VXS_VERBOSE", "1")
os.environ.pop("VXRT_SCENE_CACHE", None)
vrt = importlib.import_module("vortex-raytracing_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, args in (("atrium", ("atrium", 8, 0, 3)), ("hairball", ("hairball_fill", 20000, 250, 7))):
    if which not in (name, "both"):
        continue
    t0 = time.time()
    sc = vrt.scene.procedural(*args)
    dt = time.time() - t0
    h = hashlib.sha256()
    for k in ("tlas", "blas", "bvh", "tri"):
        h.update(np.ascontiguousarray(sc[k]).view(np.uint8).tobytes())
    print("%s: %d triangles, %d nodes, %.2f s in all (generation + build + copies into numpy), threads %s, sha %s"
          % (name, sc.n_tris, sc.n_bvh_nodes, dt, os.environ.get("VXS_THREADS", "default"), h.hexdigest()[:16]), flush=True)
