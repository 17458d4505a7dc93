#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4m; mkdir -p $O

{
for tag in base nospread; do
if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
echo "== $tag, N=8 share of 10 frames"; timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep "XCD\|^set"
done
} | tee $O/xcd.txt
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('   ', d.get('extras',{}).get('random_rays_mrays_s'), d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', c.get('sets_of_the_timed_steps', c['frames_per_launch_group']))"; }
{
for rep in 1 2; do
for tag in base nospread; do
if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
echo "$tag: 1 GPU 20 steps"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
echo "$tag: 1 GPU 200 steps"; run python bench.py --gpus 1 --no-cpu-baseline --random-rays 4194304
echo "$tag: 1 GPU serial"; run python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1
echo "$tag: rehearse 8"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 8
echo "$tag: rehearse 4"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 4
done
done
} 2>&1 | tee $O/ab.txt
