#!/bin/bash
# Counter passes (rocprofv3 --pmc, counters only) over a command, summed per kernel whose name contains <substr>.
# usage: tools/pmc_kernel.sh <substr> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <command ...>
set -u
SUB=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for p in "${PASSES[@]}"; do
  RAW=$(mktemp -d /tmp/pmck_XXXXXX)
  timeout -k 5 280 rocprofv3 --pmc $p --output-format csv -d "$RAW" -- "$@" > "$RAW/log.txt" 2>&1 || { echo "pass [$p] failed"; tail -3 "$RAW/log.txt"; continue; }
  grep "POOL=" "$RAW/log.txt" | head -2
  python "$ROOT/tools/pmc_kernel_sum.py" "$(find "$RAW" -name '*counter_collection.csv' | head -1)" "$SUB"
done
