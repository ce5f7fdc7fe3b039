#!/bin/bash
# builds profiles/_ab/libxeng_diag.so: the library with -DXENG_DIAGNOSTICS (the XENG_ABLATE / XENG_SLAB_* / XENG_MM_STREAMS
# switches that the shipped build compiles out).  usage: bash profiles/build_diag.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=${TMPDIR:-/tmp}/diagbuild
mkdir -p $B $R/profiles/_ab
cd $R/caltech-bifrost-dsp_amd/csrc
for f in xeng_util xcorr corracc beamform ingest slab ring xeng_bfarray; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXENG_DIAGNOSTICS -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/_ab/libxeng_diag.so $B/*.o
ls -la $R/profiles/_ab/libxeng_diag.so
