#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4o; mkdir -p $O
R=$PWD; L=$R/vortex-raytracing_amd/lib
{
for i in 1 2 3; do LD_LIBRARY_PATH=$L VORTEX_DRIVER=hip timeout -k 10 300 $L/rt_host -m proc:atrium:8 -w 1920 -h 1080 -S -L 300,480,60 -N 1500 -q -o /tmp/o.ppm -k $R/vortex-raytracing_amd/vxbin/kernel.vxbin 2>&1 | grep "frame loop"; done
python tools/config_bench.py 7 2>/dev/null | grep "^{"
} | tee $O/vx_path.txt
( timeout -k 10 900 python -m pytest tests -q -m gpu -x ) 2>&1 | tail -3
