#!/bin/bash
# The GPU-built tree (vxrt_bvh_build) under settings of VXRT_BVH_REINSERT ("iterations[:mod]"): what the tree is worth in the headline frame
# (tests/tree_quality.py: fetches per ray, serial frame rate beside the CPU builder's tree) and the build's time.
# usage: tools/gpu_tree_sweep.sh <out file under gpurun_out> <setting> ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for v in "$@"; do
  export VXRT_BVH_REINSERT=$v
  timeout -k 10 500 python tests/tree_quality.py --gpu --levels 8 --fixtures --leaf-max ${LEAFMAX:-2} 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k, v in d.items():
            if isinstance(v, dict) and 'mrays_s_serial' in v and (k.startswith('gpu') or k.startswith('cpu')): print('reinsert $v', k, {a: v[a] for a in ('node_fetches_per_ray', 'tri_fetches_per_ray', 'bytes_per_ray', 'mrays_s_serial', 'frame_node_fetches_per_ray', 'frame_tri_fetches_per_ray', 'nodes', 'depth', 'accel_info_levels_shallow_ident_ldexp', 'frame_stats') if a in v})
" >> "$OUT"
  VXRT_BVH_VERBOSE=1 timeout -k 10 100 python tools/bvh_build_profile.py 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/reinsert $v: /" >> "$OUT"
done
cat "$OUT"
