#!/usr/bin/env python3
"""A few serial 1920x1080 frames of the 1,048,576-triangle atrium on the CPU builder's tree (argv[1] = cpu) or on vxrt_bvh_build's (gpu),
for counter passes (tools/pmc_kernel.sh rt_persistent_kernel ... -- python tools/tree_frames.py cpu|gpu)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "cpu"
sc = vrt.scene.procedural("atrium", 8, 0, 3)
if which == "gpu":
    ds = vrt.tracer.DeviceScene.build_on_gpu(sc["tri"].view(np.float32).reshape(-1, 9), sc["triEx"].reshape(-1, 64), sc["mat"], sc["tex"], "cuda:0", leaf_max=int(os.environ.get("LEAFMAX", "2")))
else:
    ds = vrt.tracer.DeviceScene(sc, "cuda:0")
p = vrt.rtapi.default_shade_params()
p.light_pos[:] = (300.0, 480.0, 60.0)
px = torch.zeros((1080, 1920), dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    vrt.rtapi.render(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), 1, None, None, None, s)
torch.cuda.synchronize()
print(which, "pixels", int(px.to(torch.int64).sum().item()))
