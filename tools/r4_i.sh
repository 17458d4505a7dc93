#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache} VXRT_POOL_PERMILLE=0
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep -A10 "^set 2" | tee $O/wb.txt
