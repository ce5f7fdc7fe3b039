#!/bin/bash
# Collect PMC counters for bench.py in separate passes (no tracing domains combined with --pmc).
# usage: profiles/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export XENG_BENCH_TRACE=1     # bench.py names its legs on stderr: a fault under the profiler says where it was
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained 0 --no-blocks --no-h2d "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python3 $R/profiles/pmc_summarize.py "$OUT"
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4     # (the raw per-dispatch CSVs are tens of MB: the summary and pmc_traffic.json stay)
