#!/bin/bash
# Builds a scratch library = libxeng objects + csrc/diag_probe.hip and runs profiles/bperm_probe.py against it
# (on a GPU box: `gpurun -- bash profiles/bperm_probe.sh`; build the objects first with `make -C caltech-bifrost-dsp_amd/csrc`).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/caltech-bifrost-dsp_amd/csrc
mkdir -p $R/gpurun_out/probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c $C/diag_probe.hip -o $R/gpurun_out/probe/diag_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out/probe/libxeng_probe.so $C/xeng_util.o $C/xcorr.o $C/corracc.o $C/beamform.o \
    $C/ingest.o $C/xeng_bfarray.o $R/gpurun_out/probe/diag_probe.o
XENG_LIB=$R/gpurun_out/probe/libxeng_probe.so python3 $R/profiles/bperm_probe.py
