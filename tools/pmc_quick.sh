#!/bin/bash
# Instruction-count / wait counters of the bench step for one build: tools/pmc_quick.sh <outdir>   (VXRT_LIB_DIR selects a variant)
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/pass$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 --other-configs none "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $pmc" | tee -a "$OUT/errors.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/pass*/
grep -A22 "rt_persistent_kernel<1, 0, false, false>" "$OUT/summary.txt" | head -24
