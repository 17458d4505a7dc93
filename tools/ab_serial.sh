#!/bin/bash
# A/B builds for strictly serial frames (the vx_start path).  usage: tools/ab_serial.sh "<flags A>" "<flags B>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for flags in "$@"; do
  VXRT_EXTRA_HIPFLAGS="$flags" python -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('vortex-raytracing_amd.build'); b.build(force=True)" > /dev/null 2>&1
  echo "== flags: [$flags]"
  for rep in 1 2; do
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial Mrays/s', d['value'], 'ms', d['ms_per_step'])"
  done
done
