#!/bin/bash
# Round 5: fabric-side read / write traffic of the contraction per CorrAcc mode (profiles/lacc_pmc_probe.py), separate --pmc passes.
# usage (GPU box): bash profiles/lacc_pmc_run.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2; do
  i=0
  for set in "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITE_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/m${mode}p$i -- python3 $R/profiles/lacc_pmc_probe.py $mode 12 > $OUT/m${mode}p$i.log 2>&1 || echo "mode $mode pass $i failed" >> $OUT/fail.log
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/lacc_traffic_by_mode.txt", "w") as fh:
    for mode in (0, 1, 2):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(out + "/m%dp*/**/*counter_collection.csv" % mode, recursive=True):
            for row in csv.DictReader(open(f)):
                if "xcorr_fused_kernel" not in row["Kernel_Name"]:
                    continue
                a = agg[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
        fh.write("mode %d (%s)\n" % (mode, ("plain dumps", "assign: a = b", "add: a += b")[mode]))
        for c, (tot, n) in sorted(agg.items()):
            fh.write("   %-28s per launch %16.1f  (n=%d)\n" % (c, tot / max(n, 1), n))
print(open(out + "/lacc_traffic_by_mode.txt").read())
PY
rm -rf $OUT/m?p?
