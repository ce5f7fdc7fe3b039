#!/bin/bash
# A/B of whole bench.py runs between environment variants on ONE box, alternating: usage (GPU box):
#   bash profiles/ab_bench_env.sh <outdir-under-gpurun_out> <rounds> "VAR=val" "VAR=val2" ...     ("" = the defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
ROUNDS=$1; shift
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    ( [ -n "$v" ] && export $v; timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --no-blocks --sustained 0 > $OUT/r${r}_v${i}.json 2> $OUT/r${r}_v${i}.err )
  done
done
python3 - "$OUT" "$@" <<'PY'
import glob, json, sys, statistics
out, variants = sys.argv[1], sys.argv[2:]
keys = [("step", lambda d: d["ms_per_step"]), ("alone_us", lambda d: d["roofline_isolated_launches"]["avg_us"]),
        ("packets", lambda d: d["ingest_unpack"]["packets_to_visibilities"]["ms_per_step"]),
        ("packets_aligned", lambda d: d["ingest_unpack"]["packets_to_visibilities_payloads_on_cache_lines"]["ms_per_step"]),
        ("config5", lambda d: d["beamform"]["full_xengine_concurrent"]["ms_per_integration"]),
        ("config5_slabs", lambda d: d["beamform"]["full_xengine_concurrent"]["from_packet_slabs"]["in_place"]["ms_per_integration"]),
        ("sync_per_call", lambda d: d["sync_per_call"]["ms_per_step"])]
for i, v in enumerate(variants, 1):
    rows = []
    for f in sorted(glob.glob(out + "/r*_v%d.json" % i)):
        try:
            rows.append(json.load(open(f)))
        except Exception as e:
            print("unreadable", f, e)
    line = "%-24s" % (v or "default")
    for name, fn in keys:
        vals = []
        for d in rows:
            try:
                vals.append(fn(d))
            except Exception:
                pass
        line += "  %s %s" % (name, "/".join("%.4f" % x for x in vals))
    print(line)
PY
