import importlib, sys, os, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import torch
vrt = importlib.import_module("vortex-raytracing_amd")
rtapi = vrt.rtapi
W, H = 1920, 1080
sc = vrt.scene.procedural("atrium", 8, 0, 3)
dev = "cuda:0"
ND = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dss = [vrt.tracer.DeviceScene(sc, dev) for _ in range(ND)]
p = rtapi.default_shade_params(); p.light_pos[:] = (300.0, 480.0, 60.0)
streams = [torch.cuda.Stream(device=dev) for _ in range(ND)]
frames = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(ND)]
def step(i):
    k = i % ND
    rtapi.render(dss[k].accel, W, H, 0, H, p, frames[k].data_ptr(), 1, None, None, None, streams[k].cuda_stream)
for i in range(6): step(i)
torch.cuda.synchronize()
K = 60
t0 = time.perf_counter()
for i in range(K): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("depth", ND, "ms/step %.4f" % (dt / K * 1e3), "Mrays/s %.1f" % (4146808 * K / dt / 1e6))
