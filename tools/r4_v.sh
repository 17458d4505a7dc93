#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4v; mkdir -p $O
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
A="--no-cpu-baseline --random-rays 0 --other-configs none"
run() { "$@" 2>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', d['value'], d['ms_per_step'], 'kernel_ms', r['kernel_ms'])"; }
for rep in 1 2 3; do
  for t in "" "3,2" "2,2,1" "4,3,2,1" "2,1" "1"; do echo "--tail-sets '$t'"; run python bench.py --gpus 1 --steps 20 --warmup 5 $A --tail-sets "$t"; done
done
