#!/usr/bin/env python3
"""Is one XCD slower than another, and at what?  Runs csrc/calib_kernels.hip:vxcal_xcd_probe_kernel (lib/libvxrt_calib.so, measurement only):
every wavefront times three loops of fixed work -- dependent VALU instructions, a dependent pointer chase through a buffer far larger than
the L2, dependent device-scope atomics on one word -- on the shader clock and on the constant 100 MHz clock, and says which XCD it ran on.
usage: tools/xcd_probe.py [waves=2048] [rounds=4]     (idle machine; `--after-load` first runs 300 headline frames so that the clocks are up)"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
vrt = importlib.import_module("vortex-raytracing_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_waves = int(args[0]) if len(args) > 0 else 2048
rounds = int(args[1]) if len(args) > 1 else 4
dev = "cuda:0"
L = C.CDLL(vrt.lib_path("libvxrt_calib.so"))
L.vxcal_xcd_probe.restype = C.c_int
L.vxcal_xcd_probe.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
n = 1 << 26                                              # 256 MB of u32: past L2 and Infinity Cache
perm = torch.randperm(n, device=dev, dtype=torch.int32)
chase = torch.empty(n, dtype=torch.int32, device=dev)
chase[perm.long()] = torch.roll(perm, 1)                 # one random cycle over the whole buffer
del perm
atom = torch.zeros(64, dtype=torch.int32, device=dev)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
if "--after-load" in sys.argv:
    sc = vrt.scene.procedural("atrium", 8, 0, 3)
    ds = vrt.tracer.DeviceScene(sc, dev)
    p = vrt.rtapi.default_shade_params(); p.light_pos[:] = (300.0, 480.0, 60.0)
    px = torch.zeros((1080, 1920), dtype=torch.int32, device=dev)
    for _ in range(300):
        vrt.rtapi.render(ds.accel, 1920, 1080, 0, 1080, p, px.data_ptr(), 1, None, None, None, torch.cuda.current_stream().cuda_stream)
N_ALU, N_CHASE, N_ATOM = 2000, 300, 300
for rnd in range(rounds):
    out = torch.zeros((n_waves, 8), dtype=torch.int64, device=dev)
    assert L.vxcal_xcd_probe(n_waves, N_ALU, chase.data_ptr(), n, N_CHASE, atom.data_ptr(), N_ATOM, out.data_ptr(), sink.data_ptr(), None) == 0
    torch.cuda.synchronize()
    raw = out.cpu().numpy()
    o = raw.astype(np.float64)
    xcd = (raw[:, 0] & 0xFF).astype(np.int64)
    print("round %d: %d wavefronts; per XCD: wavefronts | VALU loop (%d dependent v_fma_f32 per lane): us, shader clocks per instruction, GHz seen | pointer chase ns per load, shader clocks | atomic ns per add"
          % (rnd, n_waves, 64 * N_ALU))
    for x in range(8):
        m = xcd == x
        if not m.any():
            continue
        alu_us, alu_clk = o[m, 2] / 100.0, o[m, 3]
        print("   XCD %d: %4d | %.1f us (p10 %.1f p90 %.1f), %.3f clocks per instruction, %.3f GHz | %.0f ns (p10 %.0f p90 %.0f), %.0f clocks | %.0f ns (p10 %.0f p90 %.0f)" %
              (x, int(m.sum()), np.median(alu_us), *np.percentile(alu_us, (10, 90)), np.median(alu_clk) / (64.0 * N_ALU), np.median(alu_clk / np.maximum(o[m, 2], 1) / 10.0),
               np.median(o[m, 4]) * 10.0 / N_CHASE, *(np.percentile(o[m, 4], (10, 90)) * 10.0 / N_CHASE), np.median(o[m, 5]) / N_CHASE,
               np.median(o[m, 6]) * 10.0 / N_ATOM, *(np.percentile(o[m, 6], (10, 90)) * 10.0 / N_ATOM)))
