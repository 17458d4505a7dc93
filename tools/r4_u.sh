#!/bin/bash
# the driver's command (20 steps, 5 warmup): where the time beyond 20 x the steady-state step goes
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4u; mkdir -p $O
export VXRT_SCENE_CACHE=/tmp/vxrt_scene_cache
A="--no-cpu-baseline --random-rays 0 --other-configs none"
run() { "$@" 2>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', d['value'], d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'B', d['config']['frames_per_launch_group'])"; grep "step starts" $O/err.txt | cut -c1-300; }
for rep in 1 2; do
  echo "default 20/5"; VXRT_BENCH_TRACE=1 run python bench.py --gpus 1 --steps 20 --warmup 5 $A
  for b in 4 10 20; do echo "--batch $b"; VXRT_BENCH_TRACE=1 run python bench.py --gpus 1 --steps 20 --warmup 5 $A --batch $b; done
  for f in 3 4; do echo "--frames-in-flight $f"; VXRT_BENCH_TRACE=1 run python bench.py --gpus 1 --steps 20 --warmup 5 $A --frames-in-flight $f; done
  echo "200 steps"; run python bench.py --steps 200 --warmup 20 $A
done
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 $A > $O/tr.log 2>&1
k=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/pipeline_timeline.py $k 60 > $O/timeline.txt; rm -rf $O/tr
tail -45 $O/timeline.txt
