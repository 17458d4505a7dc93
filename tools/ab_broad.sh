#!/bin/bash
# variants across the configurations: default bench (+16 Mi random rays), serial frames, driver command, the other BASELINE configs
cd ${GRAFT_REPO_ROOT:-/root/repo}
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
  echo "== $tag"
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  default', d['value'], 'random16M', d['extras'].get('random_rays_mrays_s'))"
  python bench.py --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  serial', d['value'])"
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  20/5', d['value'])"
  python bench.py --no-cpu-baseline --random-rays 0 --rehearse-world 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  rehearse8 20/5 ms', d['ms_per_step'])"
  python tools/config_bench.py 2 3 4 5 7 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'][:60], d.get('mrays_s', d.get('mrays_s_start_wait')), d.get('ms_per_frame', d.get('ms_per_frame_serial', d.get('ms_per_frame_start_wait'))))"
done
