#!/usr/bin/env python3
"""profiles/valu_profile.json from a tools/pmc_passes.sh summary: the per-frame constants bench.py's roofline uses.

VALU roof.  One bench step = main traversal launch + EXACT launches + shading pass.  Their wave64 VALU instructions per frame
(SQ_INSTS_VALU and its class counters; the counts are deterministic for a given frame) are priced with the SIMD cycles per
instruction measured by tools/calibrate_valu.py on this chip (profiles/r02_valu_calibration.txt):
    ADD_F32 / MUL_F32 / FMA_F32 (v_add, v_sub, v_mul, v_fma, v_fmac) ............ 2.2 cycles
    TRANS_F32 (v_rcp, v_sqrt) ..................................................... 8.1
    CVT (v_cvt_f32_ubyteN) ......................................................... 4.1
    INT32 (v_add_u32 / v_and 2.2; v_lshlrev / v_lshl_add / v_mul_lo 4.1) .......... 3.0 (bounds 2.2 .. 4.1)
    unclassified rest (v_cmp, v_cndmask, v_min/v_max/v_max3 4.1; v_mov 2.2) ....... 3.6 (bounds 2.2 .. 4.1)
The central estimate and both bounds are written; bench.py divides by the live kernel time.
HBM traffic: FETCH_SIZE / WRITE_SIZE (KiB, separate passes); reads x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request of a
wide read stream: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact.
usage: tools/roofline_from_pmc.py <summary.txt> <out.json> [source label]"""
import json
import re
import sys

summary, out = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else summary
cur, data = None, {}
for line in open(summary):
    if line.startswith("void ") or line.startswith("accel_"):
        cur = line.strip()
        data.setdefault(cur, {})
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m and cur:
            data[cur][m.group(1)] = float(m.group(3))
            data[cur]["_n_" + m.group(1)] = int(m.group(2))
# kernels of one timed step (STATS = 0 instantiations; the counting builds run once outside the timed region)
step = {k: v for k, v in data.items() if re.search(r"rt_persistent_kernel<1, (0|false), ", k) or "rt_shade_kernel<false>" in k}
main = next(k for k in step if re.search(r"<1, (0|false), false, false>", k))
n_main = step[main]["_n_SQ_INSTS_VALU"]


def per_frame(counter):   # launches per frame = dispatches of the kernel / dispatches of the main kernel
    tot = 0.0
    for k, v in step.items():
        if counter in v:
            tot += v[counter] * v["_n_" + counter] / n_main
    return tot


cls = {c: per_frame("SQ_INSTS_VALU_" + c) for c in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
total = per_frame("SQ_INSTS_VALU")
rest = total - sum(cls.values())
full = cls["ADD_F32"] + cls["MUL_F32"] + cls["FMA_F32"]


def cycles(p_int, p_rest):
    return 2.2 * full + 8.1 * cls["TRANS_F32"] + 4.1 * cls["CVT"] + p_int * cls["INT32"] + p_rest * rest


rd = 2.0 * per_frame("FETCH_SIZE") * 1024
wr = per_frame("WRITE_SIZE") * 1024
res = {
    "source": label,
    "valu_instr_per_frame": int(total),
    "valu_instr_classes": {k: int(v) for k, v in cls.items()} | {"unclassified": int(rest)},
    "valu_simd_cycles_per_frame": int(cycles(3.0, 3.6)),
    "valu_simd_cycles_bounds": [int(cycles(2.2, 2.2)), int(cycles(4.1, 4.1))],
    "pricing": "cycles per wave64 instruction per SIMD (tools/calibrate_valu.py, profiles/r02_valu_calibration.txt): add/mul/fma 2.2, trans 8.1, cvt 4.1, "
               "int32 3.0 (2.2..4.1), unclassified (cmp/cndmask/min/max 4.1, mov 2.2) 3.6 (2.2..4.1)",
    "hbm_bytes_per_frame": int(rd + wr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
    "hbm_source": label + " (FETCH_SIZE x2 + WRITE_SIZE, separate passes, serial frames)",
    "main_kernel": {k: v for k, v in step[main].items() if not k.startswith("_n_")},
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("valu_instr_per_frame", "valu_instr_classes", "valu_simd_cycles_per_frame", "valu_simd_cycles_bounds", "hbm_bytes_per_frame")}))
