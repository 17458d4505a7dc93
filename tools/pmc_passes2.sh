#!/bin/bash
# memory-path counters for the render kernels, few counters per pass (TA/TCP/TD have 2 slots).
# usage: tools/pmc_passes2.sh <outdir> [bench args]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for pmc in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  echo "pass $i: $pmc"
  timeout -k 5 150 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/pass$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $pmc" | tee -a "$OUT/errors.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
