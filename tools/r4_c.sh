#!/bin/bash
# round 4: bands path -- GPU tests of the changed files, then rank 0's pipeline rehearsed with the new split
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4c; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_sharding_gloo.py -x -q -m gpu --durations=15 ) > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -30 $O/tests.log
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, sets', c.get('sets_of_the_timed_steps', c['frames_per_launch_group']), 'plan', c.get('band_plan'))"; }
{
echo "1 GPU, 20 steps:"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
for n in 8 4 2; do
  echo "rehearse $n bands, taper auto:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n
  echo "rehearse $n bands, 10+10:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --taper 10,10
  echo "rehearse $n bands, one set of 20:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --taper 20
  echo "rehearse $n bands, 12,6,2:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --taper 12,6,2
  echo "rehearse $n tilerows (round 3):"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --shard tilerows
done
echo "rehearse 8 bands, 200 steps:"; run python bench.py --steps 200 --warmup 20 --no-cpu-baseline --random-rays 0 --rehearse-world 8
} 2>&1 | tee $O/rehearse.txt
cd /tmp && export TMPDIR=/tmp
for v in "8 auto" "8 10,10"; do
  set -- $v
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/t -- python $R/bench.py --no-cpu-baseline --random-rays 0 --steps 20 --warmup 5 --rehearse-world $1 --taper $2 > $R/$O/n$1_$2.log 2>&1
  python $R/tools/pipeline_timeline.py $(ls $R/$O/t/*/*kernel_trace.csv $R/$O/t/*kernel_trace.csv 2>/dev/null | head -1) 60 > $R/$O/n$1_$2_timeline.txt
  rm -rf $R/$O/t
done
echo done
