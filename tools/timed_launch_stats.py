#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `python bench.py ...`: the launches of the TIMED region only.  The whole-run averages of
--stats mix them with the untimed ones (clock-settling frames, the single-frame launches of the isolated measurement, the counting
builds), so the last `groups` traversal launches -- the timed region is the last thing bench.py renders -- are averaged here, with
the shading launches that belong to them; this is what roofline.launch_set_ms_overlapped of the same run must agree with.
usage: tools/timed_launch_stats.py <kernel_trace.csv> <groups>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
groups = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
# main launch of a frame job: rt_persistent_kernel<JOB 0|1, STATS 0, LDEXP, EXACT false, PACKED any>
main = [r for r in rows if re.search(r"rt_persistent_kernel<[01], 0, (true|false), false(, (true|false))*>", r["Kernel_Name"])]
shade = [r for r in rows if "rt_shade_kernel<false>" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
m, s = main[-groups:], shade[-groups:]
print("launches in the trace: traversal %d, shading %d; timed region = the last %d of each" % (len(main), len(shade), groups))
print("traversal launch (rt_persistent_kernel, main): avg %.1f us, min %.1f, max %.1f" % (sum(map(dur, m)) / len(m), min(map(dur, m)), max(map(dur, m))))
print("shading launch (rt_shade_kernel): avg %.1f us" % (sum(map(dur, s)) / len(s)))
span = (int(s[-1]["End_Timestamp"]) - int(m[0]["Start_Timestamp"])) / 1e3
print("span first traversal start -> last shading end: %.1f us = %.1f us per launch set" % (span, span / groups))
sets = [(int(b["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(m, s)]
print("per launch set (traversal start -> its shading end, overlapped with the other stream's set): avg %.1f us" % (sum(sets) / len(sets)))
