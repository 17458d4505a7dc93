#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4k; mkdir -p $O
for tag in base nodry; do
if [ "$tag" = "base" ]; then unset VXRT_LIB_DIR; else export VXRT_LIB_DIR=$PWD/vortex-raytracing_amd/lib_ab/$tag; fi
echo "== $tag"; timeout -k 10 200 python tools/wave_balance_batch.py 8 10 0 2>&1 | grep -A11 "^set 2"
done | tee $O/wb.txt
unset VXRT_LIB_DIR
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('   ', d.get('extras',{}).get('random_rays_mrays_s'), d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', c.get('sets_of_the_timed_steps', c['frames_per_launch_group']))"; }
{
for rep in 1 2; do
for v in "VXRT_POOL_PERMILLE=0" "VXRT_POOL_PERMILLE=250" "VXRT_LPT=0 VXRT_LPT_BATCH=0"; do
echo "$v: 1 GPU serial"; env $v python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
echo "$v: rehearse 8 10+10"; env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 8 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
echo "$v: rehearse 4 10+10"; env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 4 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
done
done
} 2>&1 | tee $O/ab.txt
