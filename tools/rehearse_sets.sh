#!/bin/bash
# Rank 0's pipeline of an N-GPU run rehearsed on ONE GPU (everything a rank does except the collective) for different schedules of the 20 timed
# steps: set sizes (--sets), sets in flight.  usage: tools/rehearse_sets.sh <world> "<sets> <sets> ..." [frames-in-flight ...]
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=$1; SETS=$2; shift 2
FIF=${@:-2}
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('   %.4f ms/step  sets %s  in flight %d  wire %s (%d bytes per rank in the last set)' % (d['ms_per_step'], c.get('sets_of_the_timed_steps'), c['frames_in_flight'], c.get('wire_format'), c.get('wire_bytes_per_rank_last_set', 0)))"; }
for rep in 1 2; do
  for f in $FIF; do
    for s in $SETS; do
      run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --random-rays 0 --other-configs none --rehearse-world $N --sets $s --frames-in-flight $f
    done
  done
done
