#!/bin/bash
# A/B several builds of the HIP library on one GPU box in one call.
# usage: tools/ab_bench.sh "<flags A>" "<flags B>" ...   (flags passed through VXRT_EXTRA_HIPFLAGS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for flags in "$@"; do
  VXRT_EXTRA_HIPFLAGS="$flags" python -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('vortex-raytracing_amd.build'); b.build(force=True)" > /dev/null 2>&1
  echo "== flags: [$flags]"
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --random-rays 4194304 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], d['extras'])"
done
