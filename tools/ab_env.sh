#!/bin/bash
# A/B of an environment knob on one GPU box, alternating runs: driver command (20 steps) and a 200-step run per setting.
# usage: tools/ab_env.sh <outfile> <reps> VAR=val1 VAR=val2 ...     (e.g. VXRT_FUSED=1 VXRT_FUSED=0)
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; REPS=$2; shift 2
mkdir -p "$(dirname "$OUT")"
for rep in $(seq $REPS); do
  for kv in "$@"; do
    env $kv python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv  driver cmd (20 steps):', d['value'], d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
    env $kv python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv  200 steps:', d['value'], d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done | tee "$OUT"
