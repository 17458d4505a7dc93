#!/bin/bash
# longest-tile-first (by work) inside larger sets: LPT_BATCH_MAX_TILES 100 K (base) / 200 K / 1 M; one GPU's 5-frame sets are 162 K tiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
O=gpurun_out/r4ad; mkdir -p $O
A="--no-cpu-baseline --other-configs none --random-rays 0"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"; }
{
for rep in 1 2; do for v in base lpt200k lpt1m; do
  if [ "$v" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$v; fi
  echo "$v: 200 steps / 20 steps / rehearse 2 / 3840x2160 serial 50 steps"
  run python bench.py $A
  run python bench.py $A --gpus 1 --steps 20 --warmup 5
  run python bench.py $A --steps 20 --warmup 5 --rehearse-world 2
  run python bench.py $A --width 3840 --height 2160 --steps 50 --warmup 5
done; done
} | tee $O/lpt_max_ab.txt
