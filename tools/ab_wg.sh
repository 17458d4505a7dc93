#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for tag in base w1 w2; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  for f in 2 3; do
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag fif $f', d['value'], d['ms_per_step'], d['config']['frames_per_launch_group'])"
  done
done
done
