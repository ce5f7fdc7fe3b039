#!/bin/bash
# Round 4: the operand-select sweep for the other VOP3P classes (v_pk_mov_b32, v_pk_add_u16, v_pk_fma_f16) beside the X-engine,
# the packed-fp32 control rows, and the stand-alone reproducer of the upstream report.  One run on a GPU box:
#   gpurun -- bash profiles/hazard/sweep_classes.sh > gpurun_out/hazard_classes.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
echo "== environment"
cat /opt/rocm/.info/version 2>/dev/null || true
/opt/rocm/bin/hipcc --version | head -2
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|Name: +gfx|Compute Unit|Max Clock" | head -8 || true
cat /sys/module/amdgpu/version 2>/dev/null || true
for f in /sys/class/drm/card*/device/fw_version/mec_fw_version /sys/class/drm/card*/device/fw_version/rlc_fw_version /sys/class/drm/card*/device/fw_version/sdma_fw_version; do [ -r $f ] && echo "$f: $(cat $f)"; done 2>/dev/null | head -6 || true
uname -r
echo "== packed fp32, the two forms of the report (control)"
bash $R/profiles/hazard/probe.sh 4 5
echo "== VOP3P classes: v_pk_mov_b32 (200-215), v_pk_add_u16 (216-231), v_pk_fma_f16 (232-247)"
XENG_LIB=$R/profiles/hazard/build/libxeng_probe.so python3 $R/profiles/hazard/probe.py 200 247
echo "== stand-alone reproducer (no libxeng)"
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $R/profiles/hazard/repro_standalone.hip -o $R/profiles/hazard/build/repro_standalone
$R/profiles/hazard/build/repro_standalone
