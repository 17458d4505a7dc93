#!/bin/bash
# A/B builder knobs on the GPU (no rebuild): tools/tree_ab.sh "VXS_LEAF_K=2.0" "VXS_LEAF_K=1.5 VXS_LEAF_MAX=8" ...
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}   # built scenes are kept between the processes of this script (keyed by the builder's knobs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do
  env $kv python bench.py --steps 100 --warmup 10 --no-cpu-baseline --random-rays 4194304 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r['counts_reference_order']; print('%-44s' % '$kv', 'Mrays/s', d['value'], 'nodes/ray %.2f tris/ray %.2f' % (c['node_fetches']/c['rays'], c['tri_fetches']/c['rays']), 'B/ray', r['bytes']['bytes_per_ray'], 'random', d['extras'].get('random_rays_mrays_s'), 'bvh nodes', d['config']['bvh_nodes'], 'depth', d['config']['bvh_depth'])"
done
