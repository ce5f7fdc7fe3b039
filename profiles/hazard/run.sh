#!/bin/bash
# Hazard lab: ONE run of profiles/integrate_beside_xengine.py per variant library (profiles/hazard/build.py), same box.
# usage (GPU box): bash profiles/hazard/run.sh <outfile> [rounds] [variants...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$1; shift
K=${1:-96}; shift
V="$@"
[ -z "$V" ] && V=$(cut -f1 $R/profiles/hazard/lib/variants.txt)
: > $OUT
for v in $V; do
  echo "== $v: $(grep -P "^$v\t" $R/profiles/hazard/lib/variants.txt | cut -f2)" >> $OUT
  XENG_LIB=$R/profiles/hazard/lib/libxeng_$v.so timeout -k 10 120 python3 $R/profiles/integrate_beside_xengine.py $K >> $OUT 2>&1 || echo "   (run failed or timed out)" >> $OUT
done
cat $OUT
