#!/usr/bin/env python3
"""VALU-roof calibration: independent streams of one vector instruction, of known length, at 1..8 wavefronts
per SIMD.  Per (instruction, occupancy): wall time per launch (HIP events), the in-kernel shader clock the chip held
(s_memtime / s_memrealtime), and the SIMD cycles one wave64 instruction occupies = shader cycles of the slowest wavefront x
1 / (instructions issued on its SIMD) -- the constant that prices SQ_INSTS_VALU of the traversal kernel.  Run it plain for the
timing, and under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` to see what the
counters read for a pipe whose utilisation is known.  usage: tools/calibrate_valu.py [n_iter] [--quick]"""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

vrt = importlib.import_module("vortex-raytracing_amd")
L = C.CDLL(vrt.lib_path("libvxrt_calib.so"))     # the measurement library (csrc/calib_kernels.hip), not the product's
L.vxcal_valu_loop.restype = C.c_int
L.vxcal_valu_loop.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
L.vxcal_instr_per_iter.restype = C.c_uint32
CUS, SIMDS = 256, 1024
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_iter = int(args[0]) if args else 4000
quick = "--quick" in sys.argv
per_iter = L.vxcal_instr_per_iter()
OPS = [(0, "v_fma_f32"), (1, "v_pk_fma_f32"), (2, "v_add_f32"), (3, "v_cndmask_b32 (vcc)"), (4, "v_cvt_f32_ubyte1"), (5, "v_max3_f32"), (6, "v_rcp_f32"),
       (7, "v_cndmask_b32_e64 (sgpr pair)"), (8, "v_mov_b32"), (9, "v_cmp_lt_f32 -> vcc"), (10, "v_cmp_lt_f32_e64 -> sgpr pair"), (11, "v_mul_f32"),
       (12, "v_max_f32"), (13, "v_and_b32"), (14, "v_cndmask_b32_e64 (sgpr pair, set once outside the loop)"),
       (15, "v_min_f32"), (16, "v_xor_b32"), (17, "v_bfi_b32"), (18, "v_sub_f32"), (19, "v_lshlrev_b32"), (20, "v_add_u32"), (21, "v_med3_f32"),
       (22, "v_fmac_f32"), (23, "v_mul_lo_u32"), (24, "v_perm_b32"), (25, "v_cndmask_b32 (vcc set by s_mov before the loop)"), (26, "v_ashrrev_i32"),
       (27, "v_min_u32"), (28, "v_lshl_or_b32"), (29, "v_and_or_b32"), (30, "v_pk_mul_f32"), (31, "v_pk_add_f32"),
       (32, "v_cndmask_b32_e64 with vcc as the explicit mask"), (33, "v_cndmask_b32 (vcc) alternating with v_add_f32"),
       (34, "v_cndmask_b32 (vcc), destination != sources"), (35, "1 v_cndmask_b32 (vcc) per 7 v_add_f32"),
       (36, "v_cvt_f32_f16"), (37, "v_fma_mix_f32 (f16 source)"), (38, "v_cvt_f32_u32"), (39, "v_cvt_f32_ubyte0"), (40, "v_lshrrev_b32"), (41, "v_bfe_u32"),
       (42, "v_add3_u32"), (43, "v_mad_u32_u24"), (44, "v_cvt_f32_u32_sdwa (byte select)"), (45, "v_cmp_lt_i32 -> vcc"), (46, "v_sub_u32"), (47, "v_min_i32"),
       (48, "MIX 1:1 v_add_f32 / v_max_f32"), (49, "MIX 1:1 v_fma_f32 / v_cvt_f32_ubyte1"), (50, "MIX 1:1 v_add_f32 / v_cndmask_b32_e64"),
       (51, "MIX 1:1 v_add_f32 / v_cmp_lt_f32_e64"), (52, "MIX 3:1 v_add_f32 / v_max_f32"), (53, "MIX 3:1 v_add_f32 / v_rcp_f32")]
if "--only-new" in sys.argv:
    OPS = [o for o in OPS if o[0] in (2, 3, 32, 33, 34, 35)]
if "--mix" in sys.argv:
    OPS = [o for o in OPS if o[0] >= 48 or o[0] in (2, 12)]
if "--set3" in sys.argv:
    OPS = [o for o in OPS if o[0] >= 36]
if quick:
    OPS = OPS[:2]
out = torch.zeros(8 * CUS * 256, dtype=torch.float32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
best = {}
for op, name in OPS:
    for wps in ((1, 2, 4, 6, 8) if op < 1 else ((1, 2, 6) if op >= 48 else (6,))):
        blocks = CUS * wps
        clocks = torch.zeros(blocks * 4 * 2, dtype=torch.int64, device="cuda:0")
        for _ in range(2):
            assert L.vxcal_valu_loop(op, blocks, n_iter, out.data_ptr(), clocks.data_ptr(), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            assert L.vxcal_valu_loop(op, blocks, n_iter, out.data_ptr(), clocks.data_ptr(), s) == 0
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ck = clocks.cpu().numpy().reshape(-1, 2).astype("float64")
        ghz = float((ck[:, 0] / ck[:, 1]).mean() * 0.1)                # shader cycles per 10 ns tick
        per_wave = per_iter * n_iter
        # a wavefront's loop lasts ck[:,0] cycles while its SIMD issues wps * per_wave instructions (if the launch spread evenly)
        cyc_inkernel = float(ck[:, 0].mean()) / (wps * per_wave)
        row = {"op": name, "waves_per_simd": wps, "ms": round(ms, 4), "wave64_instr_per_simd": wps * per_wave,
               "shader_clock_GHz": round(ghz, 3), "simd_cycles_per_instr": round(cyc_inkernel, 3),
               "Ginstr_s_chip": round(wps * per_wave * SIMDS / (ms * 1e-3) / 1e9, 1)}
        best[name] = max(best.get(name, 0.0), row["Ginstr_s_chip"])
        print(json.dumps(row), flush=True)
print(json.dumps({"peak_Ginstr_s_chip": best,
                  "reading": "a wave64 VALU instruction occupies its SIMD for simd_cycles_per_instr cycles once >= 2 wavefronts share the SIMD; "
                             "v_pk_fma_f32 performs two lane operations in the same slot"}))
