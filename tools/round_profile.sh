#!/bin/bash
# Round-end evidence in one GPU call: GPU tests, rocprofv3 kernel stats of the default (pipelined) and of the serial bench
# command, PMC passes -> profiles/valu_profile.json constants, then the default bench line (which reads those constants) and
# the serial one.  usage: tools/round_profile.sh <tag>
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
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt"; rm -rf "$OUT"/pmc/pass*/
python tools/roofline_from_pmc.py "$OUT/pmc_summary.txt" "$ROOT/profiles/valu_profile.json" "profiles/${TAG}_pmc.txt" > "$OUT/valu_profile.log" 2>&1
cp "$ROOT/profiles/valu_profile.json" "$OUT/valu_profile.json"
python tools/calibrate_valu.py 2000 --quick > "$OUT/calib_quick.txt" 2>&1
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
python bench.py --frames-in-flight 1 --no-cpu-baseline > "$OUT/bench_serial.json" 2>> "$OUT/bench.err" || exit 1
cat "$OUT/bench_serial.json" | cut -c1-400
for b in 0 1 0 1 0 1; do python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames per set of launches', d['config']['frames_per_launch_group'], 'frames in flight', d['config']['frames_in_flight'], d['value'], d['ms_per_step'])"; done > "$OUT/fif.txt" 2>&1
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command (--steps 20 --warmup 5):', d['value'], d['ms_per_step'])"; done >> "$OUT/fif.txt" 2>&1
cat "$OUT/fif.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_pipelined" -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/stats_pipelined.log" 2>&1 || { echo "rocprof pipelined failed"; exit 1; }
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- python "$ROOT/bench.py" --no-cpu-baseline --frames-in-flight 1 > "$OUT/stats_serial.log" 2>&1 || { echo "rocprof serial failed"; exit 1; }
for m in pipelined serial; do
  f=$(find "$OUT/stats_$m" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$m.csv" && head -8 "$OUT/kernel_stats_$m.csv"
  if [ "$m" = "pipelined" ]; then k=$(find "$OUT/stats_$m" -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python "$ROOT/tools/timed_launch_stats.py" "$k" 40 > "$OUT/timed_launch_stats.txt" 2>&1; grep -o '"launch_set_ms_overlapped": [0-9.]*' "$OUT/stats_pipelined.log" >> "$OUT/timed_launch_stats.txt"; cat "$OUT/timed_launch_stats.txt"; fi
  if [ "$m" = "serial" ]; then k=$(find "$OUT/stats_$m" -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python "$ROOT/tools/exact_timeline.py" "$k" > "$OUT/exact_timeline.txt" 2>&1; fi
  rm -rf "$OUT/stats_$m"
done
cat "$OUT/exact_timeline.txt"
