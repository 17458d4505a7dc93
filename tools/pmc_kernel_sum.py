#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 counter_collection.csv for kernels whose name contains argv[2]: the LAST dispatch of each such kernel."""
import collections, csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by = collections.defaultdict(float)
name = {}
for r in rows:
    d = int(r["Dispatch_Id"])
    by[(d, r["Counter_Name"])] += float(r["Counter_Value"])
    name[d] = re.sub(r"\s+", "", r["Kernel_Name"])[:70]
last = {}
for d in sorted(name):
    last[name[d]] = d
for k, d in last.items():
    vals = {c: v for (dd, c), v in by.items() if dd == d}
    if max(vals.values()) < 1e5:
        continue
    print("  %s (dispatch %d): " % (k, d) + "  ".join("%s %.4g" % (c, v) for c, v in sorted(vals.items())))
