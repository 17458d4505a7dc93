#!/bin/bash
# Per-XCD hardware counters of the headline frame's traversal launch: one rocprofv3 --pmc pass per base counter, each split by XCC through
# derived counters (tools/xcd_counters.yaml: reduce(select(C,[DIMENSION_XCC=[k]]),sum)).  Counter passes only (no tracing options).
# usage: tools/xcd_counters.sh <outdir> [base counters...]      (raw profiler output stays under /tmp; <outdir> gets summary.txt and the logs)
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
if [ $# -gt 0 ]; then BASES="$*"; else BASES="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE TCC_HIT TCC_MISS"; fi
cd /tmp && export TMPDIR=/tmp
for b in $BASES; do
  pmc=""; for k in 0 1 2 3 4 5 6 7; do pmc="$pmc ${b}_XCC$k"; done
  RAW=$(mktemp -d /tmp/xcdc_XXXXXX)
  timeout -k 5 200 rocprofv3 -E "$ROOT/tools/xcd_counters.yaml" --pmc $pmc --output-format csv -d "$RAW" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 --other-configs none --settle-frames 20 > "$OUT/$b.log" 2>&1 || { echo "pass $b failed"; tail -3 "$OUT/$b.log"; continue; }
  f=$(find "$RAW" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/tools/xcd_counters_summary.py" "$f" "$b" | tee -a "$OUT/summary.txt"
done
