#!/bin/bash
# A/B on one box, alternating: the occlusion ray's start reuses the primary direction from the lane's context (base) or derives it again (lib_ab/noreuse)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
O=gpurun_out/r4ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
A="--no-cpu-baseline --other-configs none --random-rays 0"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"; }
{
for rep in 1 2 3; do for v in base noreuse; do
  if [ "$v" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$v; fi
  echo "$v: 200 steps / 20 steps / serial"
  run python bench.py $A
  run python bench.py $A --gpus 1 --steps 20 --warmup 5
  run python bench.py $A --frames-in-flight 1
done; done
} | tee $O/reuse_ab.txt
