#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4d; mkdir -p $O
{
echo "== N=8, 10 frames per set"; timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0
echo "== N=8, 20 frames per set"; timeout -k 10 200 python tools/wave_balance_batch.py 8 20 0
echo "== N=8, 10 frames, LPT off"; VXRT_LPT_BATCH=0 timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0
echo "== N=1 (whole frames), 5 per set"; timeout -k 10 200 python tools/wave_balance_batch.py 1 5 0
} > $O/wave_balance_batch.txt 2>&1
cat $O/wave_balance_batch.txt | grep -v Warning
