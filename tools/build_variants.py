#!/usr/bin/env python3
"""Prebuild kernel variants for an A/B on the GPU box (hipcc cross-compiles here; box time is for measuring).

usage: tools/build_variants.py tag1="-DRT_TOP_NODES=240 -DRT_WG_WAVES=12" tag2="..."
Each variant lands in vortex-raytracing_amd/lib_ab/<tag>/ (libvortex-hip.so built with the extra flags, the host libraries
copied); select one with VXRT_LIB_DIR=<that dir> (tools/ab_variants.sh does)."""
import importlib
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bld = importlib.import_module("vortex-raytracing_amd.build")


def main():
    bld.build()
    out_root = os.path.join(bld.HERE, "lib_ab")
    procs = []
    for arg in sys.argv[1:]:
        tag, flags = arg.split("=", 1)
        d = os.path.join(out_root, tag)
        os.makedirs(d, exist_ok=True)
        for f in ("libvortex.so", "libvxrt_scene.so", "libvxrt_calib.so"):
            shutil.copy2(os.path.join(bld.LIB, f), os.path.join(d, f))
        src = [os.path.join(bld.CSRC, f) for f in bld.PRODUCT_HIP_SOURCES]
        cmd = [bld.HIPCC] + bld.HIP_FLAGS + flags.split() + ["-shared", "-o", os.path.join(d, "libvortex-hip.so")] + src
        open(os.path.join(d, "FLAGS"), "w").write(flags + "\n")
        procs.append((tag, subprocess.Popen(cmd)))
        if len(procs) % 4 == 0:
            for _, p in procs[-4:]:
                p.wait()
    for tag, p in procs:
        if p.wait() != 0:
            raise SystemExit("variant %s failed to build" % tag)
        print("built", tag)


if __name__ == "__main__":
    main()
