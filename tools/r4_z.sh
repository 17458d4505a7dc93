#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=$PWD/gpurun_out/r4z; mkdir -p $O
{
for tag in base rcw8s5 rcw8s4 rcw7s9; do
if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
echo "== $tag"; python tools/config_bench.py 6 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_frame','mrays_s','ms_per_frame_2_in_flight','mrays_s_2_in_flight','host_build_s')})"
done
} | tee $O/twin.txt
