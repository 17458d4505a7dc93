#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('driver command:', d['value'], d['ms_per_step'], 'frac', r['frac'], 'clock', r['clock_probe_ghz_before_after'], 'kernel_ms', r['kernel_ms'])"; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full_line.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/full_line.json').read()); print(d['value'], d['roofline']['frac'], d['roofline_random_rays']['frac'], [o['mrays_s'] for o in d['extras']['other_configs']], d['cpu_baseline']['value'])"
timeout -k 10 300 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu 2>&1 | tail -2
