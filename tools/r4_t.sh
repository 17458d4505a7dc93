#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4t
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
timeout -k 10 400 python tools/sorted_rays_probe.py > gpurun_out/r4t/sorted_rays.txt 2>&1; cat gpurun_out/r4t/sorted_rays.txt
timeout -k 10 200 python tools/config_bench.py 2 2>/dev/null | cut -c1-300
