#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4r; mkdir -p $O
{
for kv in "VXS_OPTIMIZE_LOCAL=2" "VXS_OPTIMIZE_LOCAL=0"; do
  export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache_$(echo $kv | tr ' =.:,' '_____')
  for rep in 1 2; do
  env $kv python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 4194304 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r['counts_timed_traversal']; print('%-30s' % '$kv', 'Mrays/s', d['value'], 'node steps', c['node_fetches'], 'random', d['extras'].get('random_rays_mrays_s'), 'nodes', d['config']['bvh_nodes'])"
  env $kv python bench.py --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('      serial', d['value'])"
  done
done
export VXRT_SCENE_CACHE=
VXS_VERBOSE=1 python -c "
import importlib, sys, time
sys.path.insert(0, '.')
vrt = importlib.import_module('vortex-raytracing_amd')
t=time.time(); sc = vrt.scene.procedural('atrium', 8, 0, 3); print('atrium 1M: generation + build s', round(time.time()-t,2))
t=time.time(); sc = vrt.scene.procedural('hairball_fill', 20000, 250, 7); print('hairball 10M: generation + build s', round(time.time()-t,2), sc.n_tris, sc.n_bvh_nodes)
" 2>&1 | grep "scene_builder\]  \|generation"
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache_cfg
python tools/config_bench.py 5 2>/dev/null | grep "^{" | cut -c1-420
} 2>&1 | tee $O/builder_ab4.txt
