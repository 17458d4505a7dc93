#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4b; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --settle-frames 0 --no-cpu-baseline --random-rays 0 > /dev/null 2>&1   # scene cache
cd /tmp && export TMPDIR=/tmp
for v in "8 0" "8 5" "8 20" "4 0"; do
  set -- $v
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/t -- python $R/bench.py --no-cpu-baseline --random-rays 0 --steps 20 --warmup 5 --rehearse-world $1 --batch $2 > $R/$O/n$1b$2.log 2>&1
  python $R/tools/pipeline_timeline.py $(ls $R/$O/t/*/*kernel_trace.csv $R/$O/t/*kernel_trace.csv 2>/dev/null | head -1) 70 > $R/$O/n$1b$2_timeline.txt
  rm -rf $R/$O/t
done
echo done
