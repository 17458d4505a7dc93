#!/bin/bash
# alternating A/B of prebuilt variants with the default bench (frames in groups): tools/ab_quick.sh [-r reps] tag...
cd ${GRAFT_REPO_ROOT:-/root/repo}
REPS=3
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
for rep in $(seq $REPS); do
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --random-rays 4194304 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], 'random', d['extras'].get('random_rays_mrays_s'))"
done
done
