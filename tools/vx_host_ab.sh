#!/bin/bash
# The drop-in call sequence from the compiled host (lib/rt_host: vx_upload_bytes + vx_start + vx_ready_wait [+ vx_copy_from_dev] per frame) under
# environment variants, alternately, on one box.  usage: tools/vx_host_ab.sh <spp> <frames> <rounds> "NAME=VAL ..." ["NAME=VAL ..." ...]   ("" = defaults)
set -u
SPP=$1; N=$2; ROUNDS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$ROOT/vortex-raytracing_amd/lib
export LD_LIBRARY_PATH=$LIB:${LD_LIBRARY_PATH:-} VORTEX_DRIVER=hip
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    out=$(env $v "$LIB/rt_host" -m proc:atrium:8 -w 1920 -h 1080 -S -L 300,480,60 -s "$SPP" -N "$N" -q -o /tmp/vx_host_ab.ppm -k "$ROOT/vortex-raytracing_amd/vxbin/kernel.vxbin" 2>&1 | grep "frame loop" | sed 's/.*): //' | tr '\n' ' ')
    echo "spp $SPP [${v:-defaults}]: $out"
  done
done
