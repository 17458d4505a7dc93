#!/bin/bash
# Round-end evidence in one GPU call: GPU tests, PMC passes -> profiles/valu_profile.json constants, the default bench line (which
# reads them) and the serial one, the driver's command, rocprofv3 kernel stats + timelines of the pipelined and the serial command,
# the other configurations and the VALU figure of their kernels.  usage: tools/round_profile.sh <tag>
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1 || { tail -5 "$OUT/pytest_gpu.txt"; exit 1; }
tail -1 "$OUT/pytest_gpu.txt"
"$ROOT/tools/pmc_passes.sh" "$OUT/pmc" > "$OUT/pmc.log" 2>&1
tail -3 "$OUT/pmc.log"
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt"; cp "$OUT/pmc/summary_rr.txt" "$OUT/pmc_summary_random_rays.txt"; rm -rf "$OUT"/pmc/pass*/
# (the 4th argument: a bench line of the counter passes' own session -- its roofline.identity block ties the constants to the tree and the kernel source)
python tools/roofline_from_pmc.py "$OUT/pmc_summary.txt" "$ROOT/profiles/valu_profile.json" "profiles/${TAG}_pmc.txt" "$OUT/pmc/pass1.log" "$OUT/pmc/pass_rr.log" "$OUT/pmc/summary_rr.txt" > "$OUT/valu_profile.log" 2>&1
cp "$ROOT/profiles/valu_profile.json" "$OUT/valu_profile.json"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cut -c1-600 "$OUT/bench.json"
python bench.py --frames-in-flight 1 --no-cpu-baseline > "$OUT/bench_serial.json" 2>> "$OUT/bench.err" || exit 1
cut -c1-300 "$OUT/bench_serial.json"
for i in 1 2 3 4 5; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command (--gpus 1 --steps 20 --warmup 5):', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'clock', d['roofline']['clock_ghz_held'])"; done > "$OUT/driver_cmd.txt" 2>&1
cat "$OUT/driver_cmd.txt"
tools/rehearse_multi.sh > "$OUT/multi_gpu_rehearsal.txt" 2>&1; tail -4 "$OUT/multi_gpu_rehearsal.txt"
python tools/config_bench.py 2 3 4 5 6 7 8 2>/dev/null | grep "^{" > "$OUT/configs.jsonl"; cut -c1-260 "$OUT/configs.jsonl"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_pipelined" -- python "$ROOT/bench.py" --no-cpu-baseline --other-configs none > "$OUT/stats_pipelined.log" 2>&1 || { echo "rocprof pipelined failed"; exit 1; }
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- python "$ROOT/bench.py" --no-cpu-baseline --frames-in-flight 1 --other-configs none > "$OUT/stats_serial.log" 2>&1 || { echo "rocprof serial failed"; exit 1; }
for m in pipelined serial; do
  f=$(find "$OUT/stats_$m" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -c1-400 "$f" | head -12 > "$OUT/kernel_stats_$m.csv" && head -6 "$OUT/kernel_stats_$m.csv" | cut -c1-200
  k=$(find "$OUT/stats_$m" -name "*kernel_trace.csv" | head -1)
  if [ "$m" = "pipelined" ]; then [ -n "$k" ] && python "$ROOT/tools/timed_launch_stats.py" "$k" 40 > "$OUT/timed_launch_stats.txt" 2>&1; grep -o '"launch_set_ms_overlapped": [0-9.]*' "$OUT/stats_pipelined.log" >> "$OUT/timed_launch_stats.txt"; python "$ROOT/tools/pipeline_timeline.py" "$k" 40 > "$OUT/pipeline_timeline.txt" 2>&1; cat "$OUT/timed_launch_stats.txt"; fi
  if [ "$m" = "serial" ]; then [ -n "$k" ] && python "$ROOT/tools/exact_timeline.py" "$k" > "$OUT/exact_timeline.txt" 2>&1; fi
  rm -rf "$OUT/stats_$m"
done
cat "$OUT/exact_timeline.txt"
# the other traversal kernels (diffuse bounce, hairball AO, software twin): durations from a trace, instruction counts from counter passes
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ok_trace" -- python "$ROOT/tools/config_bench.py" 3 5 6 > "$OUT/ok_trace.log" 2>&1
timeout -k 5 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/ok_pmc1" -- python "$ROOT/tools/config_bench.py" 3 5 6 > "$OUT/ok_pmc1.log" 2>&1
timeout -k 5 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/ok_pmc2" -- python "$ROOT/tools/config_bench.py" 3 5 6 > "$OUT/ok_pmc2.log" 2>&1
python "$ROOT/tools/other_kernels_roofline.py" $(find "$OUT/ok_trace" -name "*kernel_trace.csv" | head -1) $(find "$OUT/ok_pmc1" "$OUT/ok_pmc2" -name "*counter_collection.csv") > "$OUT/other_kernels_roofline.txt" 2>&1
rm -rf "$OUT/ok_trace" "$OUT/ok_pmc1" "$OUT/ok_pmc2"
cat "$OUT/other_kernels_roofline.txt"
