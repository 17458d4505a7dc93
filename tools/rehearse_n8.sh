cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['ms_per_step'], 'ms/step, B', d['config']['frames_per_launch_group'], 'fif', d['config']['frames_in_flight'])"; }
for rep in 1 2; do
for b in 0 20 7; do for f in 2 1 3; do
  echo "N=8 --batch $b --frames-in-flight $f"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 8 --batch $b --frames-in-flight $f
done; done
echo "N=8 no gather (launches alone)"; VXRT_BENCH_NO_GATHER=1 run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 8
echo "N=4 b0 f2"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 4
echo "N=4 b20 f2"; run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --rehearse-world 4 --batch 20
done
