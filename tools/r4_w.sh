#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VXRT_SCENE_CACHE=${VXRT_SCENE_CACHE:-/tmp/vxrt_scene_cache}
O=$PWD/gpurun_out/r4w; mkdir -p $O
unset VXRT_LIB_DIR
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $pmc --output-format csv -d $O/pass$i -- python $GRAFT_REPO_ROOT/tools/config_bench.py 6 > $O/pass$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r4w"
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob(O+"/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rc_persistent" in r["Kernel_Name"]: acc["rc_persistent_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/pmc.txt","w") as out:
    for k,cs in acc.items():
        print(k, file=out)
        for c,v in sorted(cs.items()): print("  %-28s n=%-3d mean=%.6g"%(c,len(v),sum(v)/len(v)), file=out)
print(open(O+"/pmc.txt").read())
PY
rm -rf $O/pass*/
