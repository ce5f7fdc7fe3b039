#!/bin/bash
# builds profiles/_ab/libxeng_k16_s<N>.so: the diagnostic library with the experimental 16x16x64 eight-wave K loop
# (csrc/experiments/xcorr_fused16.h, selected per launch with XENG_KLOOP=16), one build per scheduling variant XF16_SCHED = N
# (1: 2-2-2-3 VALU pinned behind the MFMAs, 2: the compiler's own order).  usage: bash profiles/build_kloop16.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/profiles/_ab
cd $R/caltech-bifrost-dsp_amd/csrc
for sch in 1 2; do
    B=${TMPDIR:-/tmp}/k16build_$sch
    mkdir -p $B
    for f in xeng_util corracc beamform ingest slab ring xeng_bfarray; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXENG_DIAGNOSTICS -c $f.hip -o $B/$f.o &
    done
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXENG_DIAGNOSTICS -DXENG_EXPERIMENTS -DXF16_SCHED=$sch -c xcorr.hip -o $B/xcorr.o &
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_ab/libxeng_k16_s$sch.so $B/*.o
done
ls -la $R/profiles/_ab/
