"""Rules on the ISA of the shipped code objects (CPU test: disassembles the gfx950 code objects inside libxeng.so).

Rule 1 (profiles/hazard/): no packed-fp32 VOP3P instruction (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32) with op_sel:[0,1,...], i.e. whose LOW result takes the low register of src0 and the HIGH register of
src1.  On MI355X that form returns a wrong low result in lanes 48-63 whenever an MFMA kernel shares the CU (round 2:
beam_integrate_kernel's cross-power sums; isolated at the instruction level in round 3).  hipcc chooses these forms by
itself when it packs fp32 arithmetic, so the check runs on what was actually built."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "caltech-bifrost-dsp_amd", "libxeng.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# Round 4: the fault belongs to the operand SELECT (op_sel[src0] = 0, op_sel[src1] = 1), not to one opcode
# (profiles/r03/hazard_pk_opsel_sweep.txt: the same for v_pk_mul / add / fma_f32), so the rule covers every VOP3P instruction
# that carries an op_sel -- all v_pk_*, v_fma_mix*, v_dot* -- whatever the sweep of the other classes has measured
# (profiles/r04/hazard_vop3p_class_sweep.txt); today's binary contains packed fp32 only.
VOP3P = r"\b(v_pk_\w+|v_fma_mix\w*|v_mad_mix\w*|v_dot\d\w*)\b"
BAD_PK = re.compile(VOP3P + r".*\bop_sel:\[0,1[,\]]")


def code_objects(lib, tmp):
    """gfx950 code objects of a HIP shared library: the .hip_fatbin section is a sequence of clang offload bundles."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + 24)
        off = pos + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s:
                path = os.path.join(tmp, "co_%d.co" % len(out))
                with open(path, "wb") as fh:
                    fh.write(data[pos + o:pos + o + s])
                out.append(path)
        pos = data.find(magic, pos + 1)
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="llvm-objdump not available")
def test_no_packed_fp32_with_op_sel_lo_hi(tmp_path):
    assert os.path.exists(LIB), "build libxeng.so first (__graft_entry__.build())"
    cos = code_objects(LIB, str(tmp_path))
    assert len(cos) >= 3, "expected the code objects of xcorr, beamform, corracc, ingest: %d found" % len(cos)
    npk, bad, kernels = 0, [], 0
    for co in cos:
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
        sym = "?"
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                sym = m.group(1)
                kernels += 1
                continue
            if re.search(VOP3P, line):
                npk += 1
                if BAD_PK.search(line):
                    bad.append("%s: %s" % (sym, line.strip()))
    assert kernels >= 20 and npk > 0, "disassembly looks empty (%d symbols, %d VOP3P instructions)" % (kernels, npk)
    assert not bad, "VOP3P instruction with op_sel:[0,1] (wrong low result beside MFMA kernels, DESIGN.md 4.10):\n" + "\n".join(bad[:20])


def test_rule_matches_the_failing_instruction():
    """the pattern flags the instruction of the round-2 kernel and its relatives, and not the forms measured clean"""
    assert BAD_PK.search("v_pk_mul_f32 v[30:31], v[30:31], v[24:25] op_sel:[0,1]")
    assert BAD_PK.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert BAD_PK.search("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[0,1]")
    # round 4: every VOP3P class with that select, not only the three packed-fp32 opcodes
    for bad in ("v_pk_mov_b32 v[22:23], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]",
                "v_pk_fma_f16 v1, v2, v3, v4 op_sel:[0,1,0] op_sel_hi:[1,1,1]",
                "v_pk_add_u16 v1, v2, v3 op_sel:[0,1]",
                "v_pk_mul_lo_u16 v1, v2, v3 op_sel:[0,1] op_sel_hi:[0,0]",
                "v_pk_fma_bf16 v1, v2, v3, v4 op_sel:[0,1,1]",
                "v_fma_mix_f32 v1, v2, v3, v4 op_sel:[0,1,0] op_sel_hi:[1,1,0]",
                "v_dot2_f32_f16 v1, v2, v3, v4 op_sel:[0,1,0]"):
        assert BAD_PK.search(bad), bad
    for ok in ("v_pk_mov_b32 v[16:17], v[18:19], v[18:19] op_sel:[1,0]", "v_pk_add_u16 v1, v2, v3 op_sel:[1,1]", "v_pk_fma_f16 v1, v2, v3, v4",
               "v_mfma_i32_32x32x32_i8 a[0:15], v[0:3], v[4:7], a[0:15]", "v_add_f32_e32 v1, v2, v3"):
        assert not BAD_PK.search(ok), ok
    for ok in ("v_pk_mul_f32 v[30:31], v[24:25], v[30:31] op_sel:[1,0]",
               "v_pk_mul_f32 v[28:29], v[20:21], v[22:23] op_sel:[1,1] op_sel_hi:[0,1]",
               "v_pk_fma_f32 v[22:23], v[22:23], v[24:25], v[30:31] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]",
               "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]",
               "v_pk_add_f32 v[10:11], v[10:11], v[28:29]"):
        assert not BAD_PK.search(ok), ok


# ---- Rule 2 (round 5; DESIGN.md 4.7, profiles/r04/pmc_fault_diagnosis.txt): no bare hipMemset / hipMemsetD* in the library's
# sources.  hipMemset returns before its fill has run, and the library's streams are non-blocking (they do not wait for the null
# stream): a kernel enqueued right after such a call can write the buffer first and lose its words to the late fill -- the GPU
# fault of round 4 (null descriptor base under rocprofv3 --pmc).  Initialisation-time fills go through hip_memset_now
# (xeng_common.h: fill + wait); fills on a library stream use hipMemsetAsync on THAT stream.  A source rule, because a twelfth
# bare call compiles and passes every test.
CSRC = os.path.join(ROOT, "caltech-bifrost-dsp_amd", "csrc")
BARE_MEMSET = re.compile(r"\bhipMemset(?:D8|D16|D32|2D|3D)?\s*\(")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def test_no_bare_hipmemset_in_library_sources():
    bad, seen_helper = [], False
    for dirpath, _, files in os.walk(CSRC):
        for f in files:
            if not f.endswith((".hip", ".h", ".cpp", ".hpp")):
                continue
            path = os.path.join(dirpath, f)
            lines = _strip_comments(open(path).read()).split("\n")
            inside_helper = False
            for no, line in enumerate(lines, 1):
                if "hip_memset_now(void*" in line.replace(" *", "*"):
                    inside_helper, seen_helper = True, True
                if BARE_MEMSET.search(line) and not inside_helper:
                    bad.append("%s:%d: %s" % (os.path.relpath(path, ROOT), no, line.strip()))
                if inside_helper and line.startswith("}"):
                    inside_helper = False
    assert seen_helper, "hip_memset_now (xeng_common.h) not found: the rule has lost its anchor"
    assert not bad, "bare hipMemset outside hip_memset_now (returns before the fill has run; see DESIGN.md 4.7):\n" + "\n".join(bad)


def test_memset_rule_matches():
    assert BARE_MEMSET.search("    XENG_HIP(hipMemset(p, 0, n));")
    assert BARE_MEMSET.search("hipMemsetD32 (p, 0, n)")
    assert not BARE_MEMSET.search("XENG_HIP(hipMemsetAsync(dst, value, nbytes, s));")
    assert not BARE_MEMSET.search("XENG_HIP(hip_memset_now(p, 0, n));")
    assert not BARE_MEMSET.search(_strip_comments("// hipMemset(p, 0, n) returns early\nint x; /* hipMemset( */"))
