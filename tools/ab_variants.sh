#!/bin/bash
# A/B of prebuilt kernel variants (tools/build_variants.py) on one GPU box: default pipelined bench + serial bench, per tag.
# usage: tools/ab_variants.sh [-r reps] tag1 tag2 ...      (tag "base" = the in-tree lib/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
REPS=2
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  echo "== $tag: $(cat $VXRT_LIB_DIR/FLAGS 2>/dev/null)"
  for rep in $(seq $REPS); do
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  pipelined Mrays/s', d['value'], 'ms', r['kernel_ms'], 'iso', r['kernel_ms_isolated'], 'random', d['extras'].get('random_rays_mrays_s'))"
  done
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --frames-in-flight 1 --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  serial    Mrays/s', d['value'], 'ms', r['kernel_ms'])"
done
