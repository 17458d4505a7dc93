#!/bin/bash
# A/B several builds of the HIP library on one GPU box in one call, default (pipelined) bench, 2 repeats each.
# usage: tools/ab_bench.sh "<flags A>" "<flags B>" ...   (flags passed through VXRT_EXTRA_HIPFLAGS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for flags in "$@"; do
  VXRT_EXTRA_HIPFLAGS="$flags" python -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('vortex-raytracing_amd.build'); b.build(force=True)" > /dev/null 2>&1
  echo "== flags: [$flags]"
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline  2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s', d['value'], 'ms', r['kernel_ms'], 'iso', r['kernel_ms_isolated'], d['extras']['random_rays_mrays_s'])"
  done
done
