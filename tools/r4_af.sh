#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
for rep in 1 2; do for v in base rc3 rc4; do
  if [ "$v" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$v; fi
  echo "== $v (frame contexts of the twin's layout)"; python tools/twin_frames_in_flight.py 2>/dev/null
done; done
