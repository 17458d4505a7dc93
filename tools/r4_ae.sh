#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
A="--no-cpu-baseline --other-configs none --random-rays 0 --steps 20 --warmup 5"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], 'B', d['config']['frames_per_launch_group'])"; }
for rep in 1 2; do for g in 0 2 3; do for n in 8 4; do
  echo "VXRT_GRID_DIV=$g rehearse $n"; VXRT_GRID_DIV=$g run python bench.py $A --rehearse-world $n
done; done; done
for g in 2; do for f in 3 4; do echo "VXRT_GRID_DIV=$f frames-in-flight $f batch 7/5 rehearse 8"; VXRT_GRID_DIV=$f run python bench.py $A --rehearse-world 8 --frames-in-flight $f; done; done
