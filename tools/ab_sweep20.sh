#!/bin/bash
# sweeps on one GPU box for the driver's command (20 steps) and a 200-step run: frames per set of launches, frames in flight, side reserve
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for b in 0 2 3 4 7 10; do
  echo "--batch $b: 20 steps, 200 steps"
  run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --batch $b
  run python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --batch $b
done
for f in 3; do
  echo "--frames-in-flight $f: 20 steps, 200 steps"
  run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --frames-in-flight $f
  run python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight $f
done
echo "VXRT_SIDE_RESERVE=0: 20 steps, 200 steps, serial"
VXRT_SIDE_RESERVE=0 run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
VXRT_SIDE_RESERVE=0 run python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0
VXRT_SIDE_RESERVE=0 run python bench.py --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1
done
