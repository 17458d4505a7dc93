#!/usr/bin/env python3
"""Summary of one pass of tools/xcd_counters.sh: the per-XCC values of one base counter for the last main traversal launches of the run."""
import collections, csv, sys
f, base = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if "rt_persistent_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(float)
for r in rows:
    by[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
disp = sorted({d for d, _ in by})
tot = {d: sum(v for (dd, _), v in by.items() if dd == d) for d in disp}
if not tot:
    print("%s: no dispatch of rt_persistent_kernel in %s" % (base, f))
    sys.exit(0)
big = [d for d in disp if tot[d] >= 0.5 * max(tot.values())][-4:]      # the main launches (the EXACT launches are tiny)
print("%s, per XCC, main traversal launches (dispatch ids %s):" % (base, big))
for d in big:
    vals = [by.get((d, "%s_XCC%d" % (base, k)), 0.0) for k in range(8)]
    m = sum(vals) / 8.0 or 1.0
    print("   dispatch %d: " % d + "  ".join("XCC%d %.4g (%+.1f%%)" % (k, v, 100.0 * (v / m - 1.0)) for k, v in enumerate(vals)))
