#!/bin/bash
# builder settings against each other on the headline bench line and the other configurations.
# usage: tools/ab_tree_configs.sh <out file under gpurun_out> "<env A>" "<env B>" ...
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for kv in "$@"; do
  echo "== $kv" >> "$OUT"
  env $kv python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench default:', d['value'], d['ms_per_step'], 'frac', r['frac'], 'B/ray', r['bytes']['bytes_per_ray'], 'bvh nodes', d['config']['bvh_nodes'], 'random rays', d['extras'].get('random_rays_mrays_s'))" >> "$OUT"
  env $kv python bench.py --no-cpu-baseline --frames-in-flight 1 --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench serial:', d['value'], d['ms_per_step'])" >> "$OUT"
  env $kv python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command:', d['value'], d['ms_per_step'])" >> "$OUT"
  env $kv python tools/config_bench.py 2 3 5 6 2>/dev/null | grep "^{" | cut -c1-330 >> "$OUT"
done
cat "$OUT"
