#!/bin/bash
# counters of the triangle-staging variant against the base build (serial frames, one pass per counter group): vector-memory
# instructions, L1 accesses, TA busy, wait cycles.  usage: tools/tri_lds_counters.sh <outdir> <variant tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; TAG=$2
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in base $TAG; do
  if [ "$v" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$ROOT/vortex-raytracing_amd/lib_ab/$v; fi
  i=0
  for pmc in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/$v/pass$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 --other-configs none > "$OUT/$v.pass$i.log" 2>&1 || echo "pass failed"
  done
  echo "== $v"
  python "$ROOT/tools/pmc_summary.py" "$OUT/$v" | grep -A12 "rt_persistent_kernel<1, 0, false, false, false>"
  rm -rf "$OUT/$v"
done
