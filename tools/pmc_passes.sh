#!/bin/bash
# Round-end PMC set for the bench step (serial frames), one small counter group per rocprofv3 pass,
# counters only.  usage: tools/pmc_passes.sh <outdir>
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
set -u
OUT=$1; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
cd /tmp && export TMPDIR=/tmp
# the instantiation the default (pipelined) bench times, without the serial frame's tile-cost clocks: counters of THAT code
export VXRT_PACKED=1 VXRT_LPT=0
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  echo "pass $i: $pmc"
  timeout -k 5 150 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/pass$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 --other-configs none "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $pmc" | tee -a "$OUT/errors.log"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
# one more pass WITH the random-ray leg (16 Mi rays through vxrt_trace), summarised on its own (the frame kernels' per-frame figures above
# are ratios of dispatch counts and must not see its extra dispatches): the instruction count of the ray-buffer kernel's launch
echo "pass rr: SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY (random-ray leg on)"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/rr/pass1" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --other-configs none "$@" > "$OUT/pass_rr.log" 2>&1 || echo "pass rr (random rays) failed" | tee -a "$OUT/errors.log"
timeout -k 5 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/rr/pass2" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --other-configs none "$@" > "$OUT/pass_rr2.log" 2>&1 || echo "pass rr2 (random rays, lane utilisation) failed" | tee -a "$OUT/errors.log"
python "$ROOT/tools/pmc_summary.py" "$OUT/rr" > "$OUT/summary_rr.txt" 2>&1
rm -rf "$OUT/rr"
cat "$OUT/summary.txt"
