#!/bin/bash
# Diagnostic: package power (rocm-smi) while one microbenchmark variant runs for several seconds.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/profiles/microbench
hipcc -O3 --offload-arch=gfx950 kloop_proto.hip -o /tmp/kloop_proto 2>/dev/null
hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o /tmp/mfma_shape 2>/dev/null
sample() {   # "$@" = command; prints its last line and the power / sclk seen 3 s in
  "$@" > /tmp/ps.out 2>&1 &
  local p=$!
  sleep 3
  local w=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  local w2; sleep 1; w2=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  wait $p
  echo "power ${w} / ${w2} W :: $(grep -E 'TOP/s' /tmp/ps.out | tail -1)"
}
for v in 0 1 2 3 4 5 6; do sample /tmp/kloop_proto $v 6000; done
for m in 0 1; do sample /tmp/mfma_shape 2 $m 5000; done
