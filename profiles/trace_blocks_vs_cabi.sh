#!/bin/bash
# Round 5: kernel traces of config 5 through the blocks and through the C-ABI loop on ONE box: how busy is the GPU, how long do the
# kernels take in each setting (profiles/trace_gaps.py).  usage (GPU box): bash profiles/trace_blocks_vs_cabi.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/blocks -- python3 $R/profiles/config5_blocks_repeat.py 1 300 200 > $OUT/blocks_run.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/cabi -- python3 $R/profiles/corracc_modes_probe.py 2 300 10 group > $OUT/cabi_run.txt 2>&1
for w in blocks cabi; do
  f=$(ls $OUT/$w/*/*kernel_trace.csv | head -1)
  echo "== $w: $(tail -2 $OUT/${w}_run.txt | head -1)" >> $OUT/gaps.txt
  python3 $R/profiles/trace_gaps.py $f 0.5 >> $OUT/gaps.txt 2>&1
done
rm -rf $OUT/blocks $OUT/cabi
cat $OUT/gaps.txt
