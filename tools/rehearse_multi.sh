#!/bin/bash
# rank 0's pipeline of an N-GPU run rehearsed on ONE GPU (everything a rank does per frame except the collective), driver-sized runs
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, B', d['config']['frames_per_launch_group'])"; }
echo "1 GPU, 20 steps:"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none
for n in 2 4 8; do
  for b in 0 20; do
    echo "rehearse-world $n --batch $b (20 steps):"
    run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none --rehearse-world $n --batch $b
  done
done
echo "one-rank nccl group, full size, 20 steps (the collectives with one rank):"
run python bench.py --gpus 1 --one-rank-group --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none
