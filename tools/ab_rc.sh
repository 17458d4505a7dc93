#!/bin/bash
# A/B of the software twin's frame (tools/config_bench.py 6) over prebuilt variants; usage: tools/ab_rc.sh tag1 tag2 ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  python tools/config_bench.py 6 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_frame'], 'ms', d['mrays_s'], 'Mrays/s')"
done
done
