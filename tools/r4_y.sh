#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4y; mkdir -p $O
{
nproc
python tools/builder_time.py both
VXS_THREADS=4 python tools/builder_time.py both 2>&1 | grep "sha\|binary SAH tree built\|BLAS built"
} 2>&1 | grep -v "reinsertion pass" | tee $O/builder.txt
