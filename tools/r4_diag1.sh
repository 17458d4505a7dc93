#!/bin/bash
# round 4, first GPU call: where does rank 0's 20-step run at N = 8 spend its 1.40 ms (ideal 1.01)?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=gpurun_out/r4a; mkdir -p $O
echo "== wait-value probe" | tee $O/probe.txt
timeout -k 5 30 tools/probes/waitvalue_probe >> $O/probe.txt 2>&1; echo "probe rc $?" >> $O/probe.txt
cat $O/probe.txt
run() { "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, B', d['config']['frames_per_launch_group'], 'kernel_ms', r['kernel_ms'])"; }
{
echo "1 GPU, 20 steps:"; run python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0
echo "1 GPU, 200 steps:"; run python bench.py --gpus 1 --no-cpu-baseline --random-rays 0
for n in 8 4; do
  echo "rehearse $n:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n
  echo "rehearse $n, no gather:"; VXRT_BENCH_NO_GATHER=1 run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n
  echo "rehearse $n, one set of 20:"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --batch 20
  echo "rehearse $n, one set of 20, no gather:"; VXRT_BENCH_NO_GATHER=1 run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world $n --batch 20
done
echo "rehearse 8, 200 steps (long run):"; run python bench.py --steps 200 --warmup 20 --no-cpu-baseline --random-rays 0 --rehearse-world 8
} 2>&1 | tee $O/rehearse.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/$O/n8 -o run -- python $R/bench.py --no-cpu-baseline --random-rays 0 --steps 20 --warmup 5 --rehearse-world 8 > $R/$O/n8.log 2>&1
python $R/tools/pipeline_timeline.py $(ls $R/$O/n8/*kernel_trace.csv | head -1) 80 > $R/$O/n8_timeline.txt
rocprofv3 --kernel-trace -d $R/$O/n8ng -o run -- python $R/bench.py --no-cpu-baseline --random-rays 0 --steps 20 --warmup 5 --rehearse-world 8 --batch 20 > $R/$O/n8b20.log 2>&1
python $R/tools/pipeline_timeline.py $(ls $R/$O/n8ng/*kernel_trace.csv | head -1) 40 > $R/$O/n8b20_timeline.txt
rm -rf $R/$O/n8 $R/$O/n8ng
echo done
