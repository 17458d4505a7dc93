#!/usr/bin/env python3
"""Static instruction mix of the traversal kernel's regions: compile with -DRT_ISA_MARKS -S and count, between consecutive
`; RTMARK <name>` comments of one kernel, the VALU / SALU / VMEM / LDS / branch / waitcnt instructions.  Layout order is not
execution order inside a region with branches, so this is a size estimate of each region, not a cycle count.
usage: tools/isa_regions.py [kernel-name-substring] [extra hipcc flags...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = sys.argv[1] if len(sys.argv) > 1 else "ILi1ELi0ELb0ELb0"
flags = sys.argv[2:]
src = os.path.join(ROOT, "vortex-raytracing_amd", "csrc", "rt_kernels.hip")
out = os.path.join(tempfile.gettempdir(), "rt_marks.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only",
       "-DRT_ISA_MARKS", "-o", out, src] + flags
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z20rt_persistent_kernel") and sub in l and l.rstrip().split(";")[0].rstrip().endswith(":"))
region, counts, order = "prologue", collections.OrderedDict(), []
def kind(op):
    if op.startswith(("v_",)): return "valu"
    if op.startswith(("s_waitcnt",)): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_",)): return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith(("ds_",)): return "lds"
    return "other"
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"): break
    m = re.match(r";\s*RTMARK (\w+)", t)
    if m:
        region = m.group(1)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    c = counts.setdefault(region, collections.Counter())
    c[kind(op)] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
print("%-10s %6s %6s %6s %5s %5s %6s %5s" % ("region", "valu", "salu", "branch", "vmem", "lds", "wait", "scratch"))
for r, c in counts.items():
    print("%-10s %6d %6d %6d %5d %5d %6d %5d" % (r, c["valu"], c["salu"], c["branch"], c["vmem"], c["lds"], c["wait"], c["scratch"]))
