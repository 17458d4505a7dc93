#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4q; mkdir -p $O
( timeout -k 10 600 python -m pytest tests/test_rc_twin.py tests/test_gpu_boundary.py -q -m gpu -x ) 2>&1 | tail -3
{
for tag in base rcw6 rcw5 rcw4; do
if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
for v in "VXRC_WIDE=1" "VXRC_WIDE=0"; do echo "== $tag $v"; env $v python tools/config_bench.py 6 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_frame','mrays_s','ms_per_frame_2_in_flight','mrays_s_2_in_flight')})"; done
done
} | tee $O/twin.txt
