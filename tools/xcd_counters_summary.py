#!/usr/bin/env python3
"""Summary of one pass of tools/xcd_counters.sh: the per-XCC values of one base counter for the main traversal launches of the run, by
instantiation -- the timed kernel (STATS 0) and the two counting builds bench.py runs once each (STATS 1: reference order, STATS 2: the
timed traversal with fetch counters; the wave-log diagnostics use that build)."""
import collections, csv, re, sys
f, base = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if "rt_persistent_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(float)
kern = {}
for r in rows:
    d = int(r["Dispatch_Id"])
    by[(d, r["Counter_Name"])] += float(r["Counter_Value"])
    kern[d] = re.sub(r"\s+", "", r["Kernel_Name"])
disp = sorted(kern)
tot = {d: sum(v for (dd, _), v in by.items() if dd == d) for d in disp}
if not tot:
    print("%s: no dispatch of rt_persistent_kernel in %s" % (base, f))
    sys.exit(0)
big = [d for d in disp if tot[d] >= 0.2 * max(tot.values())]      # main launches (the EXACT launches are tiny)
print("%s, per XCC, main traversal launches:" % base)
seen = collections.Counter()
for d in reversed(big):
    m_ = re.search(r"rt_persistent_kernel<\(?int\)?(\d+),\(?int\)?(\d+),\(?bool\)?(\w+),\(?bool\)?(\w+),\(?bool\)?(\w+),\(?bool\)?(\w+)>", kern[d]) or re.search(r"<([^>]*)>", kern[d])
    tag = "<%s>" % ",".join(m_.groups()) if m_ else kern[d][:60]
    seen[tag] += 1
    if seen[tag] > 2:
        continue
    vals = [by.get((d, "%s_XCC%d" % (base, k)), 0.0) for k in range(8)]
    m = sum(vals) / 8.0 or 1.0
    print("   %-34s dispatch %5d: " % (tag, d) + "  ".join("XCC%d %.4g (%+.1f%%)" % (k, v, 100.0 * (v / m - 1.0)) for k, v in enumerate(vals)))
