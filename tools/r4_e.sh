#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4e; mkdir -p $O
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log; tail -5 $O/tests.log
{
for pm in 0 150 250 350; do
echo "== N=8, 10 frames per set, pool $pm"; VXRT_POOL_PERMILLE=$pm timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep -A7 "^set 2"
done
echo "== N=1 serial frame... (1 frame per set via N=1, 1 frame)"; 
for pm in 0 250; do VXRT_POOL_PERMILLE=$pm timeout -k 10 200 python tools/wave_balance.py 2>&1 | tail -12; done
} > $O/wave_balance.txt 2>&1
cat $O/wave_balance.txt | grep -v "amdgpu.ids"
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', c.get('sets_of_the_timed_steps', c['frames_per_launch_group']))"; }
{
for pm in 0 150 250 350; do
export VXRT_POOL_PERMILLE=$pm
echo "pool $pm: 1 GPU 20 steps"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
echo "pool $pm: 1 GPU serial"; run python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1
for n in 8 4; do
echo "pool $pm: rehearse $n tilerows 10+10"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --shard tilerows
echo "pool $pm: rehearse $n tilerows 20"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --shard tilerows --batch 20
done
done
} 2>&1 | tee $O/rehearse.txt
