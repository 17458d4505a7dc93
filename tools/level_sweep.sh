#!/bin/bash
# Same camera, same frame, scenes of growing size (atrium level 3..8 = 1 K .. 1 M triangles): time per lane node step as the
# BVH outgrows L1 / L2 / Infinity Cache.  usage: tools/level_sweep.sh [levels...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lv in ${@:-3 4 5 6 7 8}; do
  for fif in 1 2; do
    python bench.py --level $lv --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight $fif 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; c = r.get('counts_timed_traversal') or r.get('counts_reference_order')
steps = c['node_fetches']; rays = c['rays']
print('level $lv fif $fif tris', d['config']['workload'].split()[2], 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'node fetches/ray %.2f tri tests/ray %.2f' % (steps / rays, c['tri_fetches'] / rays), 'ps per lane node fetch %.1f' % (d['ms_per_step'] * 1e9 / steps))"
  done
done
