#!/bin/bash
# Front-end counters of the traversal launch (instruction fetch, branches, scalar / LDS issue, LDS conflicts), serial frames, at the
# 1 M-triangle frame and at a scene that fits L1: tools/pmc_frontend.sh <outdir> [bench flags]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_STALL"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/pass$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 --other-configs none "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $pmc" | tee -a "$OUT/errors.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/pass*/
grep -A50 "rt_persistent_kernel<1, 0, false, false>" "$OUT/summary.txt" | head -52
