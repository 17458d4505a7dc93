#!/bin/bash
# A/B of the identity-instance start (VXRT_IDENT_ROOT=0 switches it off at accel build) on one box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
O=gpurun_out/r4ab; mkdir -p $O
A="--no-cpu-baseline --other-configs none"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', d['value'], d['ms_per_step'], 'random rays', d.get('extras', {}).get('random_rays_mrays_s'), 'node fetches timed', r['counts_timed_traversal']['node_fetches'])"; }
{
for rep in 1 2; do for v in 1 0; do
  echo "VXRT_IDENT_ROOT=$v: 200 steps / 20 steps / serial"
  VXRT_IDENT_ROOT=$v run python bench.py $A --gpus 1 --steps 20 --warmup 5 --random-rays 0
  VXRT_IDENT_ROOT=$v run python bench.py $A --frames-in-flight 1 --random-rays 0
done; done
} | tee $O/ident_ab.txt
