#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4g; mkdir -p $O
{
for pm in 0 250; do echo "== pool $pm"; VXRT_POOL_PERMILLE=$pm timeout -k 10 200 python tools/tile_tail.py 8 10; VXRT_POOL_PERMILLE=$pm timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep -A7 "^set 2"; done
} > $O/tile_tail.txt 2>&1
grep -v "amdgpu.ids" $O/tile_tail.txt | grep -v "start decile"
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', c.get('sets_of_the_timed_steps', c['frames_per_launch_group']))"; }
{
for pm in 0 250; do
export VXRT_POOL_PERMILLE=$pm
echo "pool $pm: 1 GPU 20 steps"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
echo "pool $pm: 1 GPU 200 steps"; run python bench.py --gpus 1 --no-cpu-baseline --random-rays 0
echo "pool $pm: 1 GPU serial"; run python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1
for n in 8 4; do
echo "pool $pm: rehearse $n tilerows 10+10"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --shard tilerows
echo "pool $pm: rehearse $n tilerows 20"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --shard tilerows --batch 20
done
done
} 2>&1 | tee $O/rehearse.txt
