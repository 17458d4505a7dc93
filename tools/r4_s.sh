#!/bin/bash
# multi-device backend tests + bench with the other-configs legs + full GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4s
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r4s/boundary.txt 2>&1; tail -5 gpurun_out/r4s/boundary.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r4s/bench.json 2> gpurun_out/r4s/bench.err || tail -5 gpurun_out/r4s/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4s/bench.json').read())
print(d['value'], d['ms_per_step'])
for o in d['extras'].get('other_configs', []): print(o)
PY
# the C++ host's frame loop on 1 and on 2 shares of the same card
L=vortex-raytracing_amd/lib
for devs in "" "0,0"; do
  echo "VORTEX_HIP_DEVICES='$devs'"
  LD_LIBRARY_PATH=$L VORTEX_DRIVER=hip VORTEX_HIP_DEVICES=$devs timeout -k 10 200 $L/rt_host -m proc:atrium:8 -w 1920 -h 1080 -S -L 300,480,60 -N 500 -q -o /tmp/o.ppm -k vortex-raytracing_amd/vxbin/kernel.vxbin 2>&1 | grep "frame loop"
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4s/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4s/pytest_gpu.txt
