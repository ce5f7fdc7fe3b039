#!/bin/bash
# rocprofv3 kernel-trace statistics of the shipped binary: (1) default bench with every side leg (streaming call mode),
# (2) one contraction at a time.  usage (GPU box): bash profiles/kernel_stats_run.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/streaming -- python3 $R/bench.py --no-cpu-baseline --sustained 0 > $OUT/bench_under_rocprof_streaming.json 2> $OUT/streaming.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sync -- python3 $R/bench.py --no-cpu-baseline --no-h2d --no-beamform --sustained 0 --sync-per-integration --steps 500 --warmup 50 --prewarm 300 > $OUT/bench_under_rocprof_sync_per_integration.json 2> $OUT/sync.err
cp $OUT/streaming/*/*kernel_stats.csv $OUT/kernel_stats_streaming_all_legs.csv
cp $OUT/sync/*/*kernel_stats.csv $OUT/kernel_stats_sync_per_integration.csv
rm -rf $OUT/streaming $OUT/sync
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
for name in ("kernel_stats_streaming_all_legs.csv", "kernel_stats_sync_per_integration.csv"):
    print(name)
    for r in csv.DictReader(open(out + "/" + name)):
        if float(r["Percentage"]) > 0.05 or "xeng" in r["Name"]:
            print("   %-44s calls %6s avg %10.1f ns min %9s max %9s" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
for name in ("bench_under_rocprof_streaming.json", "bench_under_rocprof_sync_per_integration.json"):
    d = json.load(open(out + "/" + name))
    print(name, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us_overlapped"))
PY
