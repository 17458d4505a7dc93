#!/usr/bin/env python3
"""When do the EXACT launches of a serial frame run relative to its main traversal launch?  Reads a rocprofv3 --kernel-trace CSV
(kernel_trace.csv) of `bench.py --frames-in-flight 1` and prints, per frame (one main launch), the start / end of the a-priori
EXACT launch (side stream), of the deferred-list EXACT launch and of the shading pass, in microseconds from the main launch's
start, plus the gap from the end of the frame's last traversal kernel to the shading pass.
usage: tools/exact_timeline.py <kernel_trace.csv> [frames to skip]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "rt_persistent_kernel<1, 0" in n or "rt_shade_kernel<false>" in n or "lpt_order_kernel" in n:
        m = re.search(r"rt_persistent_kernel<1, 0, (true|false), (true|false)", n)      # <JOB, STATS, LDEXP, EXACT, PACKED>
        kind = "shade" if "rt_shade" in n else ("lpt" if "lpt_order" in n else ("exact" if (m and m.group(2) == "true") else "main"))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "") + "/" + r.get("Stream_Id", "")))
ev.sort()
main_q = next(q for _, _, k, q in ev if k == "main")
mains = [(s, e) for s, e, k, q in ev if k == "main"]
frames = []
for i, (m0, m1) in enumerate(mains[:-1]):
    nxt = mains[i + 1][0]
    prv = mains[i - 1][1] if i else 0
    f = {"main": (m0, m1), "next": nxt}
    # a-priori EXACT launch: on the side stream (another queue), submitted just before this frame's main launch
    ap = [(s, e) for s, e, k, q in ev if k == "exact" and q != main_q and prv <= s < m1]
    df = [(s, e) for s, e, k, q in ev if k == "exact" and q == main_q and m1 <= s < nxt]
    sh = [(s, e) for s, e, k, q in ev if k == "shade" and m1 <= s < nxt]
    lp = [(s, e) for s, e, k, q in ev if k == "lpt" and m1 <= s < nxt]
    f["ap"] = ap[-1] if ap else None
    f["df"] = df[0] if df else None
    f["shade"] = sh[0] if sh else None
    f["lpt"] = lp[0] if lp else None
    frames.append(f)
frames = [f for f in frames[skip:] if f["shade"]]
import statistics as st
us = lambda a, b: (a - b) / 1e3
keys = ("main_dur", "ap_start", "ap_end", "df_start", "df_end", "lpt_start", "lpt_end", "shade_start", "shade_end", "next_main_start")
stats = {k: [] for k in keys}
for f in frames:
    m0, m1 = f["main"]
    stats["main_dur"].append(us(m1, m0))
    for tag in ("ap", "df", "lpt"):
        if f[tag]:
            stats[tag + "_start"].append(us(f[tag][0], m0)); stats[tag + "_end"].append(us(f[tag][1], m0))
    stats["shade_start"].append(us(f["shade"][0], m0)); stats["shade_end"].append(us(f["shade"][1], m0))
    stats["next_main_start"].append(us(f["next"], m0))
print("frames analysed: %d   (times in us from the start of the frame's main traversal launch; serial frames)" % len(frames))
for k in keys:
    v = stats[k]
    if v:
        print("%-16s mean %8.1f  p50 %8.1f  p90 %8.1f  max %8.1f   (n=%d)" % (k, st.mean(v), st.median(v), sorted(v)[int(0.9 * len(v))], max(v), len(v)))
late = sum(1 for f in frames if f["ap"] and f["ap"][1] > f["main"][1])
print("frames whose a-priori EXACT launch (side stream) ended AFTER the main launch: %d of %d" % (late, sum(1 for f in frames if f["ap"])))
