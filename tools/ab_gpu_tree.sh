#!/bin/bash
# prebuilt library variants (tools/build_variants.py) x the GPU-built tree: tree quality harness per variant.  usage: tools/ab_gpu_tree.sh <out> tag...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$ROOT/vortex-raytracing_amd/lib_ab/$tag; fi
  timeout -k 10 500 python tests/tree_quality.py --gpu --levels 8 --fixtures --leaf-max 2 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k, v in d.items():
            if isinstance(v, dict) and k.startswith('gpu'): print('$tag', k, {a: v[a] for a in ('node_fetches_per_ray', 'tri_fetches_per_ray', 'bytes_per_ray', 'mrays_s_serial', 'nodes', 'depth') if a in v})
" >> "$OUT"
  timeout -k 10 100 python tools/bvh_build_profile.py 2>/dev/null | tail -1 | sed "s/^/$tag /" >> "$OUT"
done
cat "$OUT"
