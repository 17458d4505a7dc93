#!/bin/bash
# usage: tools/ab_bench2.sh "<bench args>" "<flags A>" "<flags B>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
for flags in "$@"; do
  VXRT_EXTRA_HIPFLAGS="$flags" python -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('vortex-raytracing_amd.build'); b.build(force=True)" > /dev/null 2>&1
  echo "== flags: [$flags] args: [$ARGS]"
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'rays', d['config']['rays_per_step_per_gpu'], d.get('extras'))"
done
