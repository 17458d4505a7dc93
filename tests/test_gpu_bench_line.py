"""GPU: the driver's contract on bench.py -- one JSON line with BASELINE.json's metric, the roofline object (a fraction of a roof,
not above 1) and the cpu_baseline object, for the command the driver runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_driver_command_prints_one_well_formed_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2", "--random-rays", "1048576"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" at ")[0] == base["metric"].split(" at ")[0] and "1920x1080" in d["metric"]
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    assert "1048576 tris" in d["config"]["workload"] and "1920x1080" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 5000, "Mrays/s on an MI355X"
    rf = d["roofline"]
    assert rf["bound"] == "valu"
    # one consistent denominator: instructions per second against 1,024 SIMDs x the clock HELD in the timed region / 2 cycles
    assert 1.2 < rf["clock_ghz_held"] < 2.7 and all(1.2 < c < 2.7 for c in rf["clock_probe_ghz_before_after"])
    assert abs(rf["peak"] - 1024 * rf["clock_ghz_held"] / 2.0) < 0.1
    # the line says which tree, which kernel source and which traversal its numerator would have to belong to ...
    ident = rf["identity"]
    assert ident["width"] == 1920 and ident["height"] == 1080 and ident["bvh_nodes"] == d["config"]["bvh_nodes"] and len(ident["tree_sha16"]) == 16
    assert len(ident["kernel_source_sha16"]) == 16 and ident["rays"] == d["config"]["rays_per_step"] and ident["node_fetches_timed"] > 0
    prof = json.load(open(os.path.join(ROOT, "profiles", "valu_profile.json")))
    sys.path.insert(0, ROOT)
    import bench
    if bench.profile_mismatch(prof, ident) is None:      # (equal, the counting build's two fetch totals to 0.1 %)
        # ... and the checked-in counters are this run's: a fraction, recomputable from the line
        assert 0.3 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        assert abs(rf["achieved"] - rf["valu_instr_per_frame"] / (rf["kernel_ms"] * 1e-3) / 1e9) / rf["achieved"] < 1e-3
        alt = rf["same_numerator_other_denominators"]
        assert alt["frac_at_2.2_cycles_per_instruction_measured_on_this_chip"] > rf["frac"] and 0.3 < alt["frac_at_nominal_2.4_GHz_and_2_cycles"] <= 1.0
        mk = rf["main_kernel_counters"]
        assert 0.3 < mk["lane_utilisation"] <= 1.0 and 0.0 < mk["wait_any_of_wave_cycles"] < 1.0
        assert rf["traffic"] and rf["traffic"] < rf["bytes"]["algorithmic_bytes_per_launch"]      # measured HBM bytes per frame: the scene is cache-resident
    else:
        # ... or they are another tree's / another kernel's: no fraction, and the line says why (never a stale numerator)
        assert rf["frac"] is None and rf["achieved"] is None and "valu_profile.json" in rf["frac_is_null_because"]
    assert d["config"]["frames_rendered_before_the_timed_region"] >= d["warmup"] + d["config"]["clock_settle_frames_untimed"]
    assert rf["counts_timed_traversal"]["rays"] == d["config"]["rays_per_step"] == rf["counts_reference_order"]["rays"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and 0 < cb["value"] < 1000 and cb["unit"] == "Mrays/s" and cb["sample"]
    assert d["extras"]["random_rays_mrays_s"] > 500


def _bench(args, timeout=900, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_plain_command_with_two_gpus_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver starts --gpus 1) must be a 2-rank run: bench.py starts
    its ranks as child processes before touching HIP.  Here the two ranks share the box's one GPU and gather over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = _bench(["--gpus", "2", "--dist-backend", "gloo", "--steps", "6", "--warmup", "2", "--settle-frames", "4", "--level", "4",
                "--no-cpu-baseline", "--random-rays", "65536"], timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and len(d["config"]["rank_devices"]) == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["steps"] == 6 and d["value"] > 0 and d["config"]["rays_per_step"] > d["config"]["rays_per_step_rank0"] > 0


def test_more_gpus_than_the_box_has_is_an_error_not_a_one_gpu_line():
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = _bench(["--gpus", str(n + 7), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--random-rays", "0"], timeout=300, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")], r.stdout[-500:]
    assert "device" in r.stderr


_NCCL_ONE_RANK = r"""
import importlib, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
sh = importlib.import_module("vortex-raytracing_amd.sharding")
h, w, B = 1080, 1920, 3
ig = sh.InterleavedGather(h, w, 0, 1, "cuda:0", slots=2, batch=B, single_rank_collective=True)
fb = ig.new_frame_buffer("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(7)
fb.copy_(torch.randint(0, 1 << 24, fb.shape, generator=g, device="cuda:0", dtype=torch.int32))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):          # as bench.py does: the assembly runs on its own stream
    out = ig.gather(fb, 1)
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    objs = [None]
    dist.all_gather_object(objs, "cuda:0")
side.synchronize()
assert out.shape == (B, h, w) and torch.equal(out, fb[:, :h]) and float(t.item()) == 1.5 and objs == ["cuda:0"]
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_rccl_gather_of_the_image_assembly_runs_with_a_one_rank_group():
    """The N > 1 run's collectives (dist.gather of the shares on device tensors, the all_reduce of the timing, the barrier) on the
    nccl = RCCL backend with a group of ONE rank: the first RCCL call ever made by this code must not be on the driver's 8-GPU node."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    r = subprocess.run([sys.executable, "-c", _NCCL_ONE_RANK % (ROOT, port)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_the_multi_rank_code_path_of_bench_runs_over_rccl_with_a_one_rank_group():
    """bench.py's whole N > 1 path -- nccl process group on the device, interleaved batches, dist.gather on its own stream, barriers,
    all_reduce of the timing -- with a group of one rank on the box's GPU (--one-rank-group)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = _bench(["--gpus", "1", "--one-rank-group", "--steps", "20", "--warmup", "5", "--settle-frames", "10", "--level", "5",
                "--no-cpu-baseline", "--random-rays", "65536"], timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["dist_backend"] == "nccl" and d["config"]["world_size"] == 1
    assert d["config"]["frames_per_launch_group"] == 10 and "interleaved" in d["config"]["parallelism"] and d["value"] > 0
