"""CPU: build-time guards on the generated code of the traversal kernels (hipcc -S for gfx950, no GPU needed).

Round 2 lost a GPU to this: written as `s_n == 0` on a select, the "queue shard lies past the end of the job range" test of the
persistent kernel's fetch section was folded by the optimiser into a form that is true for every shard past the end, the EXACT
launch's wavefronts ran past the deferral list and the launch faulted.  The test is now an asm statement (s_cmp_lt_u32 + marker) the
optimiser cannot look into; this check makes sure every instantiation still carries it, ahead of its first queue atomic."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def listing():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = os.path.join(tempfile.gettempdir(), "vxrt_guard_listing.s")
    src = os.path.join(ROOT, "vortex-raytracing_amd", "csrc", "rt_kernels.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def _kernels(lines):
    """{mangled name: body lines} of every rt_persistent_kernel instantiation."""
    out, name = {}, None
    for l in lines:
        m = re.match(r"(_Z20rt_persistent_kernel\w+):", l)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            out[name].append(l)
            if l.strip().startswith("s_endpgm"):
                name = None
    return out


def test_every_instantiation_tests_the_shard_range_before_its_first_queue_atomic(listing):
    ks = _kernels(listing)
    assert len(ks) >= 32, "instantiations of rt_persistent_kernel in the listing: %d" % len(ks)
    for name, body in ks.items():
        guards = [i for i, l in enumerate(body) if "RTGUARD shard_range" in l]
        atomics = [i for i, l in enumerate(body) if re.match(r"\s*(global|flat)_atomic_add\b", l)]
        assert len(guards) >= 1, "%s lost the shard range test" % name
        assert atomics, name
        assert guards[0] < atomics[0], "%s: a queue atomic precedes the shard range test" % name
        # the marker rides on the compare itself
        g = body[guards[0]]
        prev = body[guards[0] - 1]
        assert "s_cselect_b32" in g and "s_cmp_lt_u32" in prev, (name, prev, g)


def test_the_software_twin_kernel_carries_the_same_guard():
    """rc_kernels.hip's rc_persistent_kernel reserves tiles from the same kind of sharded queue: its range test is the same asm statement,
    ahead of its queue atomic (the `s_lo < n_tiles` compare had been folded away there as well)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = os.path.join(tempfile.gettempdir(), "vxrc_guard_listing.s")
    src = os.path.join(ROOT, "vortex-raytracing_amd", "csrc", "rc_kernels.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    body, on = [], False
    for l in lines:
        if re.match(r"_ZN\S*rc_persistent_kernel\S*:", l) or re.match(r"\S*rc_persistent_kernel\S*:", l):
            on = True
        elif on:
            body.append(l)
            if l.strip().startswith("s_endpgm"):
                break
    assert body, "rc_persistent_kernel not found in the listing"
    guards = [i for i, l in enumerate(body) if "RTGUARD shard_range" in l]
    atomics = [i for i, l in enumerate(body) if re.match(r"\s*(global|flat)_atomic_add\b", l)]
    assert guards and atomics and guards[0] < atomics[0]
    assert "s_cselect_b32" in body[guards[0]] and "s_cmp_lt_u32" in body[guards[0] - 1]
