#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=$PWD/gpurun_out/r4x; mkdir -p $O
A="--no-cpu-baseline --random-rays 0 --other-configs none --steps 20 --warmup 5"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], 'B', d['config']['frames_per_launch_group'])"; }
{
for n in 8 4; do for f in 2 3 4; do for b in 0 20; do echo "rehearse $n fif $f batch $b"; run python bench.py $A --rehearse-world $n --frames-in-flight $f --batch $b; done; done; done
} | tee $O/rehearse_fif.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py -x -q -m gpu --durations=8 2>&1 | tail -15
