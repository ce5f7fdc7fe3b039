#!/bin/bash
# Builds a scratch library = libxeng objects + profiles/hazard/diag_probe.hip and runs profiles/hazard/probe.py against it
# (on a GPU box: `gpurun -- bash profiles/hazard/probe.sh`; build the objects first with `make -C caltech-bifrost-dsp_amd/csrc`).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/caltech-bifrost-dsp_amd/csrc
O=$R/profiles/hazard/build
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$C -c $R/profiles/hazard/diag_probe.hip -o $O/diag_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libxeng_probe.so $C/xeng_util.o $C/xcorr.o $C/corracc.o $C/beamform.o \
    $C/ingest.o $C/slab.o $C/ring.o $C/xeng_bfarray.o $O/diag_probe.o
XENG_LIB=$O/libxeng_probe.so python3 $R/profiles/hazard/probe.py "$@"
