#!/bin/bash
# A/B of the hairball AO frame (tools/config_bench.py 5) over prebuilt variants; usage: tools/ab_ao.sh tag1 tag2 ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  python tools/config_bench.py 5 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', {k:d[k] for k in d if k not in ('config','tris','bvh_nodes','bvh_depth','host_build_s')})"
done
