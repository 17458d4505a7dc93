#!/usr/bin/env python3
"""Print register / scratch metadata of the kernels in a hipcc -S listing: tools/kmeta.py file.s [substr]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sub not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(name[:70], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"),
          "vspill", g("vgpr_spill_count"), "sspill", g("sgpr_spill_count"), "lds", g("group_segment_fixed_size"))
