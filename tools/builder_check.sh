#!/bin/bash
# GPU builder in one call: its tests, build times, per-kernel times of six builds, tree quality against the reference's and the SAH builder's tree.
# usage: tools/builder_check.sh <out dir under gpurun_out> [leaf_max ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_bvh_builder.py -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 120 python tools/bvh_build_profile.py > "$OUT/build_times.txt" 2>&1; tail -3 "$OUT/build_times.txt"
timeout -k 10 500 python tests/tree_quality.py --gpu --levels 8 --fixtures --leaf-max ${@:-0} > "$OUT/tree_quality.txt" 2>&1
python - "$OUT/tree_quality.txt" <<'P'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        for k, v in d.items():
            if isinstance(v, dict): print(k, {a: v[a] for a in ('node_fetches_per_ray', 'tri_fetches_per_ray', 'bytes_per_ray', 'mrays_s_serial', 'nodes', 'depth') if a in v})
P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bp" -- python "$ROOT/tools/bvh_build_profile.py" > "$OUT/bp.log" 2>&1
f=$(find "$OUT/bp" -name "*kernel_stats.csv" | head -1)
python - "$f" > "$OUT/build_kernel_stats.txt" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
    print("%-62s calls %4s  total %9.1f us  avg %8.1f us  per build %8.1f us" % (name, r["Calls"], int(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 6e3))
P
rm -rf "$OUT/bp"; cat "$OUT/build_kernel_stats.txt"
