#!/bin/bash
# rank 0's pipeline of an N-GPU run rehearsed on ONE GPU (everything a rank does per frame except the collective), driver-sized runs
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', d['config'].get('sets_of_the_timed_steps', d['config']['frames_per_launch_group']))"; }
echo "1 GPU, 20 steps:"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none
for n in 2 4 8; do
  for rep in 1 2; do
    echo "rehearse-world $n (20 steps, sets as bench.py picks them):"
    run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none --rehearse-world $n --no-leg-4k
  done
done
echo "rehearse-world 8, 3840x2160 leg (extras.leg_3840x2160):"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none --rehearse-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['extras']['leg_3840x2160'])"
echo "one-rank nccl group, full size, 20 steps (the collectives with one rank):"
run python bench.py --gpus 1 --one-rank-group --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none
