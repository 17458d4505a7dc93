#!/usr/bin/env python3
"""profiles/valu_profile.json from a tools/pmc_passes.sh summary: the per-frame constants bench.py's roofline uses.

VALU roof.  One bench step = main traversal launch + EXACT launches + shading pass.  Their wave64 VALU instructions per frame
(SQ_INSTS_VALU; deterministic for a given frame) are set against the issue rate tools/calibrate_valu.py measures on this chip
(profiles/r02_valu_calibration.txt): a SIMD issues one wave64 VALU instruction per 2.2 cycles -- 1,100 G instructions/s over
the 1,024 SIMDs at the 2.35 GHz the loops held -- for v_add/sub/mul/fma/mov/and/xor streams AND for 1:1 mixes of those with
the opcodes that sustain only one per 4.1 cycles back to back (v_cmp, v_cndmask, v_min/v_max/v_max3, v_cvt_f32_ubyte,
v_lshlrev, packed f32); v_rcp/v_sqrt cost ~8-12.  So the roof of a mixed stream is
    max( all VALU x 2.2 ,  slow-class VALU x 4.1 ) + transcendental x 8
cycles per SIMD; the class counters (ADD/MUL/FMA/TRANS/INT32/CVT, the unclassified rest = cmp, cndmask, min/max, mov) bound the
slow class from above by  CVT + INT32 + rest.
HBM traffic: FETCH_SIZE / WRITE_SIZE (KiB, separate passes); reads x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request of a
wide read stream: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact.
The file also records the IDENTITY of the run the counters describe -- frame size, tree hash + node count, kernel source id, the
traversal counts of the counting build -- copied from the `roofline.identity` block of a bench line of the same session; bench.py
reports `frac: null` with the reason when its own run differs in any of them (a changed tree or kernel would otherwise leave a
stale numerator under a confident fraction).
usage: tools/roofline_from_pmc.py <summary.txt> <out.json> [source label] [bench_line.json] [bench_line_random_rays.json summary_rr.txt]"""
import json
import re
import sys

summary, out = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else summary
identity = None
if len(sys.argv) > 4:
    for line in open(sys.argv[4]):
        if line.lstrip().startswith("{"):
            identity = json.loads(line).get("roofline", {}).get("identity")
# optional: the bench line of the counter pass that ran WITH the random-ray leg (its extras.random_rays block identifies the ray buffer)
rr_line = None
if len(sys.argv) > 5:
    for line in open(sys.argv[5]):
        if line.lstrip().startswith("{"):
            rr_line = json.loads(line).get("extras", {}).get("random_rays")
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
# (the main launch: not EXACT; the instantiation the summary was taken with -- tools/pmc_passes.sh forces the PACKED one, which is what
# the default bench times)
mains = [k for k in step if re.search(r"rt_persistent_kernel<1, (0|false), false, false(, (true|false|0|1))*>", k)]   # <JOB, STATS, LDEXP, EXACT[, PACKED[, SHALLOW]]>
main = max(mains, key=lambda k: step[k].get("SQ_INSTS_VALU", 0.0) * step[k].get("_n_SQ_INSTS_VALU", 0))
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


slow_max = cls["CVT"] + cls["INT32"] + rest            # upper bound of the one-per-4.1-cycles class
cyc_all = 2.2 * total + 8.0 * cls["TRANS_F32"]
cyc_slow = 4.1 * slow_max + 8.0 * cls["TRANS_F32"]

rd = 2.0 * per_frame("FETCH_SIZE") * 1024
wr = per_frame("WRITE_SIZE") * 1024
res = {
    "source": label,
    "identity": identity,
    "valu_instr_per_frame": int(total),
    "valu_instr_classes": {k: int(v) for k, v in cls.items()} | {"unclassified": int(rest)},
    # the slow class is under half of the stream (static mix of the hot loop: ~35 %; the counter bound below counts every v_mov and
    # full-rate integer op as slow), so the binding term is the issue rate of the whole stream
    "valu_simd_cycles_per_frame": int(cyc_all),
    "valu_simd_cycles_all_x2.2": int(cyc_all), "valu_simd_cycles_slow_class_upper_bound_x4.1": int(cyc_slow),
    "valu_peak_Ginstr_s_measured": 1100.0,
    "pricing": "max(all VALU x 2.2, slow-class (cmp/cndmask/min/max/cvt/lshl: at most CVT + INT32 + unclassified) x 4.1) + TRANS x 8 SIMD cycles "
               "(tools/calibrate_valu.py, profiles/r02_valu_calibration.txt)",
    "hbm_bytes_per_frame": int(rd + wr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
    "hbm_source": label + " (FETCH_SIZE x2 + WRITE_SIZE, separate passes, serial frames)",
    "main_kernel": {k: v for k, v in step[main].items() if not k.startswith("_n_")},
}
# the ray-buffer kernel (rt_persistent_kernel<JOB_TRACE = 2, STATS = 0, ...>): instructions per launch of the random-ray leg
data_rr, cur = {}, None
if len(sys.argv) > 6:
    for line in open(sys.argv[6]):
        if line.startswith("void ") or line.startswith("accel_"):
            cur = line.strip()
            data_rr.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
            if m and cur:
                data_rr[cur][m.group(1)] = float(m.group(3))
tr = {k: v for k, v in data_rr.items() if re.search(r"rt_persistent_kernel<2, (0|false), ", k) and "SQ_INSTS_VALU" in v}
if tr and rr_line:
    # main + EXACT instantiation of one vxrt_trace call: both are launched once per call
    per_launch = sum(v["SQ_INSTS_VALU"] for v in tr.values())
    res["random_rays"] = {"n": rr_line["rays_per_gpu"], "node_fetches": rr_line["node_fetches"], "tri_fetches": rr_line["tri_fetches"],
                          "valu_instr_per_launch": int(per_launch), "source": label + " (pass with the random-ray leg: SQ_INSTS_VALU, mean over its launches)"}
    big = max(tr.values(), key=lambda v: v["SQ_INSTS_VALU"])       # the main instantiation (the EXACT launch carries a handful of rays)
    if big.get("SQ_ACTIVE_INST_VALU") and big.get("SQ_THREAD_CYCLES_VALU"):
        res["random_rays"]["lane_utilisation"] = round(big["SQ_THREAD_CYCLES_VALU"] / (64.0 * big["SQ_ACTIVE_INST_VALU"]), 4)
    if big.get("SQ_WAIT_ANY") and big.get("SQ_WAVE_CYCLES"):
        res["random_rays"]["wait_any_of_wave_cycles"] = round(big["SQ_WAIT_ANY"] / big["SQ_WAVE_CYCLES"], 4)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("valu_instr_per_frame", "valu_instr_classes", "valu_simd_cycles_per_frame", "valu_simd_cycles_all_x2.2", "valu_simd_cycles_slow_class_upper_bound_x4.1", "hbm_bytes_per_frame")}))
