#!/usr/bin/env python3
"""vxrt_bvh_build on the 1,048,576-triangle scene, a few times, for `rocprofv3 --kernel-trace --stats -- python tools/bvh_build_profile.py`."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = vrt.scene.procedural("atrium", level, 0, 3)
tri = sc["tri"].view(np.float32).reshape(-1, 9)
ex = sc["triEx"].reshape(-1, 64)
perm = np.random.default_rng(1).permutation(len(tri))
t0_tri, t0_ex = torch.from_numpy(tri[perm].copy()).cuda(), torch.from_numpy(ex[perm].copy()).cuda()
n = len(tri)
nodes = torch.zeros(2 * n * 52, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for i in range(6):
    a, b = t0_tri.clone(), t0_ex.clone()
    torch.cuda.synchronize()
    t0 = time.time()
    info = vrt.rtapi.bvh_build(a.data_ptr(), b.data_ptr(), n, nodes.data_ptr(), 2 * n, 0, 2, s)
    print("build %d: %.2f ms, %d nodes, depth %d" % (i, (time.time() - t0) * 1e3, info.n_nodes, info.max_depth), flush=True)
