"""CPU: how bench.py decides what kind of run it is, before anything touches a GPU -- `--gpus N` without a launcher starts N ranks
itself or fails; it never prints a 1-GPU line for an N-GPU request (Missing 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


def test_more_gpus_than_devices_is_refused():
    import torch
    n = torch.cuda.device_count()
    r = _bench(["--gpus", str(n + 8)])
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()


def test_launcher_world_size_must_match_the_gpus_flag():
    r = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_gpus_zero_is_refused():
    r = _bench(["--gpus", "0"])
    assert r.returncode != 0 and not r.stdout.strip()
