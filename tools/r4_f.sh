#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4f; mkdir -p $O
{
for pm in 0; do echo "== pool $pm"; VXRT_POOL_PERMILLE=$pm timeout -k 10 200 python tools/tile_tail.py 8 10; done
} > $O/tile_tail.txt 2>&1
grep -v "amdgpu.ids" $O/tile_tail.txt
