#!/bin/bash
# Round 5: kernel traces of the Corr block on device rings and of the C-ABI streaming loop on ONE box (profiles/trace_gaps.py).
# usage (GPU box): bash profiles/trace_corr_block.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in block loop; do
  python3 $R/profiles/corr_block_probe.py $w 600 > $OUT/${w}_plain.txt 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$w -- python3 $R/profiles/corr_block_probe.py $w 600 > $OUT/${w}_run.txt 2>&1
  f=$(ls $OUT/$w/*/*kernel_trace.csv | head -1)
  echo "== $w: unprofiled $(tail -1 $OUT/${w}_plain.txt); under the profiler $(tail -1 $OUT/${w}_run.txt)" >> $OUT/gaps.txt
  python3 $R/profiles/trace_gaps.py $f 0.5 >> $OUT/gaps.txt 2>&1
done
rm -rf $OUT/block $OUT/loop
cat $OUT/gaps.txt
