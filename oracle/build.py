"""Build the checker libraries (test infrastructure).  Building the checker is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False):
    so = os.path.join(HERE, "librt_oracle.so")
    csrc = [os.path.join(HERE, "rt_oracle.c"), os.path.join(HERE, "rc_oracle.c")]
    src = csrc + [os.path.join(HERE, "rt_oracle.h"), os.path.join(HERE, "rc_oracle.h")]
    if force or _newer(so, src):
        # flags are part of the oracle: no contraction, no -march=native, no fast-math
        cmd = ["gcc", "-std=c99", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-o", so] + csrc + ["-lm"]
        print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if os.path.isdir(os.path.join(REF, "sim", "simx")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "ref_build"), "REF=" + REF])
    return so


if __name__ == "__main__":
    build()
