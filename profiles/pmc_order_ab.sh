#!/bin/bash
# PMC passes of the shipped item order (profiles/pmc_run.sh) and one FETCH_SIZE pass with XENG_ITEM_ORDER=plain beside it
R=$GRAFT_REPO_ROOT
bash $R/profiles/pmc_run.sh r02_pmc2 --no-h2d --no-beamform > $R/gpurun_out/r02_pmc2.log 2>&1; tail -22 $R/gpurun_out/r02_pmc2.log
cd /tmp && export TMPDIR=/tmp XENG_ITEM_ORDER=plain
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc2_plain -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --no-beamform > /dev/null 2>&1
python3 - <<PY
import csv,glob
t=[0.0,0]
for f in glob.glob("$R/gpurun_out/r02_pmc2_plain/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "xcorr_fused_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": t[0]+=float(r["Counter_Value"]); t[1]+=1
print("plain order: FETCH_SIZE per dispatch KB", t[0]/max(t[1],1), "n", t[1])
PY
