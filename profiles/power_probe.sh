#!/bin/bash
# Diagnostic: board power / clocks as rocm-smi reports them while the default bench's timed region runs.
# usage: profiles/power_probe.sh <outfile-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
python3 $R/bench.py --steps 40000 --warmup 100 --no-cpu-baseline --no-h2d --no-beamform --sustained 0 > $OUT.bench.json 2> $OUT.bench.err &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)|Performance Level" >> $OUT.smi.txt
  echo "--" >> $OUT.smi.txt
  sleep 1
done
rocm-smi --showmaxpower --showpowerplay 2>&1 | tail -20 >> $OUT.smi.txt
wait $BP
python3 -c "
import json;d=json.load(open('$OUT.bench.json'));print(d['value'],d['ms_per_step'])" >> $OUT.smi.txt
cat $OUT.smi.txt
