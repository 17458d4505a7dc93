#!/bin/bash
# longest-tile-first inside batches of frames (learned from the context's previous batch of the same size), on / off:
# single GPU (driver command + 200 steps) and rank 0's pipeline of 2 / 4 / 8 ranks rehearsed on one GPU (driver-sized runs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, B', d['config']['frames_per_launch_group'])"; }
for rep in 1 2; do
for v in 1 0; do
  export VXRT_LPT_BATCH=$v
  echo "VXRT_LPT_BATCH=$v: 1 GPU 20 steps, 200 steps"
  run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
  run python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0
  for n in 2 4 8; do
    echo "VXRT_LPT_BATCH=$v: rehearse-world $n (20 steps)"
    run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n
  done
done
done
