#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
for i in 1 2; do
python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench serial          ', d['value'], d['ms_per_step'])"
VXRT_BENCH_COUNT_RAYS=1 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench serial + counter', d['value'], d['ms_per_step'])"
done
