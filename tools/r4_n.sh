#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4n; mkdir -p $O
( timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_reference_host.py tests/test_gpu_parity.py tests/test_rc_twin.py -q -m gpu ) > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log; tail -25 $O/tests.log
L=$PWD/vortex-raytracing_amd/lib
{
for i in 1 2; do
LD_LIBRARY_PATH=$L VORTEX_DRIVER=hip timeout -k 10 300 $L/rt_host -m proc:atrium:8 -w 1920 -h 1080 -S -L 300,480,60 -N 200 -q -o /tmp/o.ppm -k $PWD/vortex-raytracing_amd/vxbin/kernel.vxbin 2>&1 | grep -v "^PERF"
done
python tools/config_bench.py 7 2>/dev/null | grep "^{"
python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench serial', d['value'], d['ms_per_step'])"
} 2>&1 | tee $O/vx_path.txt
