#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 --kernel-trace CSV (proper CSV parsing: kernel names hold commas):
start / end / duration in us relative to the first listed dispatch, hardware queue, grid size, short kernel name.
usage: tools/pipeline_timeline.py <kernel_trace.csv> [N=60]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
def short(name):
    m = re.search(r"rt_persistent_kernel<(\d), (\d), (\w+), (\w+)[,>]", name)
    if m:
        return ("EXACT " if m.group(4) == "true" else "MAIN  ") + "job%s" % m.group(1)
    m = re.search(r"(\w+_kernel|\w+)(<|\()", name)
    return m.group(1) if m else name[:40]
print("%10s %10s %9s  %-5s %-6s %8s  %s" % ("start_us", "end_us", "dur_us", "queue", "stream", "grid", "kernel"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %10.1f %9.1f  %-5s %-6s %8s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r.get("Grid_Size_X", "?"), short(r["Kernel_Name"])))
