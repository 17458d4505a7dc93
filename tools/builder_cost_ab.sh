#!/bin/bash
# GPU builder: triangle cost of the collapse's dynamic programme x largest leaf, frame rate of the 1M-triangle scene on each tree.
# usage: tools/builder_cost_ab.sh <out file under gpurun_out>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT=$ROOT/gpurun_out/$1
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for tc in 18 36 72 104; do
  VXRT_BVH_TRI_COST=$tc timeout -k 10 400 python tests/tree_quality.py --gpu --levels 8 --fixtures --leaf-max 2 3 4 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for k, v in d.items():
            if isinstance(v, dict) and (k.startswith('gpu') or k.startswith('cpu')): print('tri cost $tc', k, {a: v[a] for a in ('node_fetches_per_ray', 'tri_fetches_per_ray', 'bytes_per_ray', 'mrays_s_serial', 'nodes', 'depth') if a in v})
" >> "$OUT"
done
cat "$OUT"
