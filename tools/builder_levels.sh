#!/bin/bash
# durations of every launch of ONE vxrt_bvh_build (the last of six) in launch order.  usage: tools/builder_levels.sh <out dir under gpurun_out>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/bt" -- python "$ROOT/tools/bvh_build_profile.py" > "$OUT/bt.log" 2>&1
f=$(find "$OUT/bt" -name "*kernel_trace.csv" | head -1)
python - "$f" > "$OUT/build_launches.txt" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last build: from the last bb_init_kernel on
last = max(i for i, r in enumerate(rows) if "bb_init_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f  grid %8s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), name))
P
rm -rf "$OUT/bt"; cat "$OUT/build_launches.txt"
