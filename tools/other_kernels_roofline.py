#!/usr/bin/env python3
"""The same VALU figure bench.py gives the headline kernel, for the other traversal kernels: from a rocprofv3 --kernel-trace CSV
(durations) and a rocprofv3 --pmc CSV (SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU per
dispatch) of the SAME command (tools/config_bench.py 3 5 6), per kernel:
    achieved = wave64 VALU instructions per dispatch / average dispatch duration      [G instr/s]
    peak     = 1,024 SIMDs x clock / 2 cycles per instruction                         (clock: argument, GHz)
    frac     = achieved / peak;  lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU);  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES
Counters serialise the dispatches, the trace does not: durations come from the trace run.
usage: tools/other_kernels_roofline.py <kernel_trace.csv> <counter_collection.csv>... [--clock 2.4]"""
import csv, re, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
clock = float(sys.argv[sys.argv.index("--clock") + 1]) if "--clock" in sys.argv else 2.4
trace, pmcs = args[0], args[1:]


def short(n):
    m = re.search(r"rt_persistent_kernel<(\d), (\d), (true|false), (true|false)(, (true|false))?(, (true|false))?>", n)   # <JOB, STATS, LDEXP, EXACT, PACKED, SHALLOW>
    if m:
        return None if m.group(4) == "true" else "rt_persistent_kernel<%s, %s, %s, false, %s>" % (m.group(1), m.group(2), m.group(3), m.group(6) or "false")   # (EXACT launches: a handful of rays)
    if "rc_persistent_kernel" in n:
        return "rc_persistent_kernel"
    return None


dur = defaultdict(list)
for r in csv.DictReader(open(trace)):
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = defaultdict(lambda: defaultdict(list))
for f in pmcs:
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
JOBS = {"0": "frame, primary rays", "1": "frame, primary + occlusion rays", "2": "ray buffer (vxrt_trace)", "3": "frame, primary + one diffuse bounce in the lane",
        "4": "ray buffer of any-hit rays in slot order (AO rays, a bounce level's occlusion rays)"}
print("peak = 1024 SIMDs x %.2f GHz / 2 = %.1f G wave64 VALU instructions/s" % (clock, 1024 * clock / 2))
for k in sorted(dur):
    c = cnt.get(k)
    if not c or "SQ_INSTS_VALU" not in c:
        continue
    # only the launches that carry work (the EXACT launches and empty ones would drag the averages)
    big = [d for d in dur[k] if d > 0.25 * max(dur[k])]
    iv = [v for v in c["SQ_INSTS_VALU"] if v > 0.25 * max(c["SQ_INSTS_VALU"])]
    d_us, n_valu = sum(big) / len(big), sum(iv) / len(iv)
    ach = n_valu / (d_us * 1e-6) / 1e9
    line = "%-52s launches %3d  avg %8.1f us  VALU %.4g instr/launch  achieved %6.1f G/s  frac %.3f" % (k, len(big), d_us, n_valu, ach, ach / (1024 * clock / 2))
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c:
        sel = lambda name: [v for v in c[name] if v > 0.25 * max(c[name])]
        a, t = sel("SQ_ACTIVE_INST_VALU"), sel("SQ_THREAD_CYCLES_VALU")
        line += "  lanes %.2f" % ((sum(t) / len(t)) / (64.0 * sum(a) / len(a)))
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        sel = lambda name: [v for v in c[name] if v > 0.25 * max(c[name])]
        w, y = sel("SQ_WAIT_ANY"), sel("SQ_WAVE_CYCLES")
        line += "  wait %.2f" % ((sum(w) / len(w)) / (sum(y) / len(y)))
    m = re.match(r"rt_persistent_kernel<(\d)", k)
    print(line + ("   [" + JOBS.get(m.group(1), "") + "]" if m else "   [software twin]"))
