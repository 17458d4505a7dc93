#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4l; mkdir -p $O
{
for rot in 0 0 3; do echo "== rot $rot, N=8 share of 10 frames"; VXRT_SHARD_ROT=$rot timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep "XCD\|^set 2"; done
echo "== rot 0, whole frames x5"; timeout -k 10 200 python tools/wave_balance_batch.py 1 5 0 2>&1 | grep -A30 "^set 2" | grep "XCD\|^set 2"
} | tee $O/xcd.txt
