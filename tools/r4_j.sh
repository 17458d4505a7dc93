#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4j; mkdir -p $O
( timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=8 ) > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -60 $O/tests.log
