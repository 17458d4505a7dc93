#!/bin/bash
# Throughput of the traversal kernel against resident wavefronts per SIMD (VXRT_WGS_PER_CU caps the persistent grid; one
# 4-wave workgroup per CU = one wavefront per SIMD).  Linear growth = each wavefront is bound by its own latency chain;
# saturation = the issue ports are.  usage: tools/occupancy_sweep.sh [level]
cd ${GRAFT_REPO_ROOT:-/root/repo}
LV=${1:-8}
for n in 1 2 3 4 5 6; do
  VXRT_WGS_PER_CU=$n python bench.py --level $LV --steps 60 --warmup 6 --no-cpu-baseline --random-rays 0 --frames-in-flight 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('level $LV waves/SIMD $n ms %.4f Mrays/s %.1f per-wave-slot Mrays/s %.1f' % (d['ms_per_step'], d['value'], d['value'] / $n))"
done
