#!/bin/bash
# A/B builder knobs on the GPU (no rebuild): tools/tree_ab.sh "VXS_LEAF_K=2.0" "VXS_LEAF_K=1.5 VXS_LEAF_MAX=8" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --steps 100 --warmup 10 --no-cpu-baseline --random-rays 4194304 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms', d['roofline']['kernel_ms'], 'B/ray', d['roofline']['bytes_per_ray'], 'frac', d['roofline']['frac'], d['extras'])"
done
