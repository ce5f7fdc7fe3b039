#!/bin/bash
# builds variants of the diagnostic library around xcorr_fused16_kernel (csrc/xcorr_fused16.h) for A/B runs with profiles/ab_step.py:
#   profiles/_ab/libxeng_k16_<name>.so   with the macros given as name=flags pairs, e.g.
#   bash profiles/build_kloop16.sh s1="-DXF16_SCHED=1" s3="-DXF16_SCHED=3" p1="-DXF16_PRIO=1"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/profiles/_ab
cd $R/caltech-bifrost-dsp_amd/csrc
B0=${TMPDIR:-/tmp}/k16build_common
mkdir -p $B0
for f in xeng_util corracc beamform ingest slab ring xeng_bfarray; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXENG_DIAGNOSTICS -c $f.hip -o $B0/$f.o &
done
wait
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXENG_DIAGNOSTICS $flags -c xcorr.hip -o $B0/xcorr_$name.o &
done
wait
for spec in "$@"; do
    name=${spec%%=*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_ab/libxeng_k16_$name.so $B0/xeng_util.o $B0/corracc.o $B0/beamform.o $B0/ingest.o $B0/slab.o $B0/ring.o $B0/xeng_bfarray.o $B0/xcorr_$name.o
done
ls -la $R/profiles/_ab/libxeng_k16_*.so
