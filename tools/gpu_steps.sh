#!/bin/bash
# Runs a list of GPU steps in one gpurun call, each under its own time limit, each with its own log under gpurun_out/<tag>/.
# A step that fails with an ordinary error is reported and the next one runs; a step that is KILLED at its limit (or by a signal) ends the
# call: nothing more is started on a GPU that may be hung.
# usage: tools/gpu_steps.sh <tag> <<'EOF'
#   <seconds> <name> <command ...>
#   ...
# EOF
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
cd "$ROOT"
rc_all=0
while read -r secs name cmd; do
  [ -z "${secs:-}" ] && continue
  case "$secs" in \#*) continue;; esac
  echo "=== $name (limit ${secs}s): $cmd"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "    rc $rc, $(( $(date +%s) - t0 )) s; tail:"
  tail -4 "$OUT/$name.log" | cut -c1-400 | sed 's/^/    | /'
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "step $name was killed (rc $rc): stopping here"; exit 1; fi
  [ $rc -ne 0 ] && rc_all=1
done
exit $rc_all
