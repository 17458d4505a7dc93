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
    assert rf["bound"] == "valu" and 0.3 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] and rf["traffic"] < rf["bytes"]["algorithmic_bytes_per_launch"]      # measured HBM bytes per frame: the scene is cache-resident
    assert rf["counts_timed_traversal"]["rays"] == d["config"]["rays_per_step"] == rf["counts_reference_order"]["rays"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and 0 < cb["value"] < 1000 and cb["unit"] == "Mrays/s" and cb["sample"]
    assert d["extras"]["random_rays_mrays_s"] > 500
