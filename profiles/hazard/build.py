#!/usr/bin/env python3
"""Hazard lab, DESIGN.md 4.10: instruction-level variants of the round-2 beam_integrate_kernel.

beamform_lab.hip is compiled to gfx950 assembly once; every variant below is ONE textual edit of that assembly (exact
match, exactly one occurrence, or the build stops), assembled, linked, bundled and linked with the product objects into
profiles/hazard/lib/libxeng_<variant>.so.  run.sh then runs profiles/integrate_beside_xengine.py once per library.
Needs `make -C caltech-bifrost-dsp_amd/csrc` first (the other objects).  hipcc cross-compiles: runs without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "caltech-bifrost-dsp_amd", "csrc")
LL = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + CSRC]

PK_FMA_XYR = "\tv_pk_fma_f32 v[28:29], v[22:23], v[24:25], v[30:31]\n"
PK_FMA_XYI = "\tv_pk_fma_f32 v[22:23], v[22:23], v[24:25], v[30:31] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
MOV_HI = "\tv_mov_b32_e32 v29, v23\n"
PK_ADD_ACC = "\tv_pk_add_f32 v[10:11], v[10:11], v[28:29]\n"
LOOP_EXIT = "\ts_andn2_b64 exec, exec, s[14:15]\n\ts_cbranch_execnz .LBB0_6\n"
FIRST_BPERM = "\tds_bpermute_b32 v14, v18, v12\n"
PK_MUL_OPSEL = "\tv_pk_mul_f32 v[30:31], v[30:31], v[24:25] op_sel:[0,1]\n"
PK_ADD_XXYY = "\tv_pk_add_f32 v[12:13], v[12:13], v[26:27]\n"

# name -> (what it isolates, [(old, new)])
VARIANTS = {
    "base": ("the compiler's code, unedited (the failing form)", []),
    "scalar_xyr": ("(i) the cross-power real term by a scalar v_fma_f32 v28 instead of the packed FMA whose high half (v29) is junk "
                   "and is overwritten by v_mov_b32 v29, v23 two instructions later",
                   [(PK_FMA_XYR, "\tv_fma_f32 v28, v22, v24, v30\n")]),
    "nop_before_mov": ("(ii) one s_nop 1 in front of v_mov_b32 v29, v23 (more distance to the packed FMA that wrote v23 and to the one that wrote v29)",
                       [(MOV_HI, "\ts_nop 1\n" + MOV_HI)]),
    "copy_before_bperm": ("(iii) the packed loop unchanged, but the four sums pass through v_mov_b32 before the first ds_bpermute "
                          "(the cross-lane reads no longer take a packed-fp32 result)",
                          [(FIRST_BPERM, "\tv_mov_b32_e32 v12, v12\n\tv_mov_b32_e32 v13, v13\n\tv_mov_b32_e32 v10, v10\n\tv_mov_b32_e32 v11, v11\n" + FIRST_BPERM)]),
    "uniform_exit": ("(iv) uniform trip count: the loop leaves by a scalar test of the same mask, EXEC is never narrowed",
                     [(LOOP_EXIT, "\ts_andn2_b64 s[2:3], exec, s[14:15]\n\ts_cmp_lg_u64 s[2:3], 0\n\ts_cbranch_scc1 .LBB0_6\n")]),
    "scalar_acc": ("the last accumulation v[10:11] += v[28:29] by two v_add_f32 (everything else packed): is it the packed WRITE of the "
                   "register the ds_bpermute reads?",
                   [(PK_ADD_ACC, "\tv_add_f32_e32 v10, v10, v28\n\tv_add_f32_e32 v11, v11, v29\n")]),
    "nop_after_acc": ("s_nop 7 x2 behind the last packed add of the loop body (timing only)",
                      [(PK_ADD_ACC, PK_ADD_ACC + "\ts_nop 7\n\ts_nop 7\n")]),
    "no_mov_scalar_acc": ("no v_mov_b32 v29, v23 at all (no write behind the packed FMA that also wrote v29): the sums take v28 and v23 "
                          "by two v_add_f32",
                          [(MOV_HI, ""), (PK_ADD_ACC, "\tv_add_f32_e32 v10, v10, v28\n\tv_add_f32_e32 v11, v11, v23\n")]),
    # ---- second series: none of the edits above clears it, so the wrong term is made before the accumulation ----
    "scalar_mul_yy": ("the products a.y*b.y | a.x*b.y by two v_mul_f32 instead of v_pk_mul_f32 ... op_sel:[0,1] (the low result reads the HIGH "
                      "register of src1)", [(PK_MUL_OPSEL, "\tv_mul_f32_e32 v30, v30, v25\n\tv_mul_f32_e32 v31, v31, v25\n")]),
    "opsel_swapped": ("the same packed multiply with its sources swapped (op_sel:[1,0]: the low result reads the high register of src0)",
                      [(PK_MUL_OPSEL, "\tv_pk_mul_f32 v[30:31], v[24:25], v[30:31] op_sel:[1,0]\n")]),
    "nop_before_opsel": ("s_nop 3 in front of the op_sel multiply (distance to the v_mov_b32 that wrote v30, v31)", [(PK_MUL_OPSEL, "\ts_nop 3\n" + PK_MUL_OPSEL)]),
    "nop_after_opsel": ("s_nop 3 behind the op_sel multiply (distance to its readers)", [(PK_MUL_OPSEL, PK_MUL_OPSEL + "\ts_nop 3\n")]),
    "xyi_elsewhere": ("the cross-power imaginary FMA writes v[26:27] instead of v[22:23] (no write to v22 right behind the instruction that reads it): "
                      "the xx|yy add moves in front of it, v29 takes v27",
                      [(PK_FMA_XYI + PK_ADD_XXYY, PK_ADD_XXYY + PK_FMA_XYI.replace("v_pk_fma_f32 v[22:23], v[22:23]", "v_pk_fma_f32 v[26:27], v[22:23]")),
                       (MOV_HI, "\tv_mov_b32_e32 v29, v27\n")]),
    "scalar_xyi": ("the cross-power imaginary term by one v_fma_f32 v23, v23, v24, -v31 (no junk write to v22 at all)",
                   [(PK_FMA_XYI, "\tv_fma_f32 v23, v23, v24, -v31\n")]),
    # ---- third series: what about that multiply?  (v32.. : the kernel's register budget is raised to 40) ----
    "dest_elsewhere": ("the op_sel multiply writes a third register pair (v[32:33]) instead of its own src0; its two readers take v[32:33]",
                       [(PK_MUL_OPSEL, "\tv_pk_mul_f32 v[32:33], v[30:31], v[24:25] op_sel:[0,1]\n"),
                        (PK_FMA_XYR, "\tv_pk_fma_f32 v[28:29], v[22:23], v[24:25], v[32:33]\n"),
                        (PK_FMA_XYI, PK_FMA_XYI.replace("v[30:31] op_sel_hi", "v[32:33] op_sel_hi")), "vgpr40"]),
    "src1_copy": ("the op_sel multiply unchanged (D = S0), but src1 is a VALU copy of the loaded pair (v[32:33] = v[24:25]): is it the register pair a "
                  "global_load has just written?",
                  [(PK_MUL_OPSEL, "\tv_mov_b32_e32 v32, v24\n\tv_mov_b32_e32 v33, v25\n\tv_pk_mul_f32 v[30:31], v[30:31], v[32:33] op_sel:[0,1]\n"), "vgpr40"]),
    "src0_copy": ("the op_sel multiply with src0 in a third pair and D = the old src0 pair: v_pk_mul_f32 v[30:31], v[32:33], v[24:25] op_sel:[0,1]",
                  [(PK_MUL_OPSEL, "\tv_mov_b32_e32 v32, v30\n\tv_mov_b32_e32 v33, v31\n\tv_pk_mul_f32 v[30:31], v[32:33], v[24:25] op_sel:[0,1]\n"), "vgpr40"]),
}


def run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


def main():
    out = os.path.join(HERE, "lib")
    tmp = os.path.join(HERE, "build")
    os.makedirs(out, exist_ok=True)
    os.makedirs(tmp, exist_ok=True)
    lab = os.path.join(HERE, "beamform_lab.hip")
    base_s = os.path.join(tmp, "lab.s")
    run([HIPCC] + FLAGS + ["--cuda-device-only", "-S", lab, "-o", base_s], stderr=subprocess.DEVNULL)
    text = open(base_s).read()
    # the loop label is numbered by the kernel's position in the file: find it
    import re
    m = re.search(r"s_andn2_b64 exec, exec, s\[14:15\]\n\ts_cbranch_execnz (\.LBB\d+_\d+)\n", text)
    assert m, "loop exit of the failing kernel not found"
    label = m.group(1)
    objs = [os.path.join(CSRC, o) for o in ("xeng_util.o", "xcorr.o", "corracc.o", "ingest.o", "slab.o", "ring.o", "xeng_bfarray.o")]
    want = sys.argv[1:] or list(VARIANTS)
    for name in want:
        why, edits = VARIANTS[name]
        t = text
        for e in edits:
            if e == "vgpr40":      # raise the kernel's VGPR budget (kernel descriptor + metadata) so that v32..v39 exist
                assert t.count(".amdhsa_next_free_vgpr 32") == 1 and t.count(".vgpr_count:     32") == 1, "vgpr budget lines"
                assert t.count(".amdhsa_accum_offset 32") == 1
                t = t.replace(".amdhsa_next_free_vgpr 32", ".amdhsa_next_free_vgpr 40").replace(".vgpr_count:     32", ".vgpr_count:     40")
                t = t.replace(".amdhsa_accum_offset 32", ".amdhsa_accum_offset 40")
                continue
            old, new = e
            old, new = old.replace(".LBB0_6", label), new.replace(".LBB0_6", label)
            assert t.count(old) == 1, "%s: %d matches of %r" % (name, t.count(old), old)
            t = t.replace(old, new)
        s = os.path.join(tmp, name + ".s")
        open(s, "w").write(t)
        o, co, fb, ho = (os.path.join(tmp, name + e) for e in (".o", ".out", ".hipfb", ".host.o"))
        run([LL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
        run([LL + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", co, o])
        run([LL + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + co, "-output=" + fb])
        run([HIPCC] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", lab, "-o", ho],
            stderr=subprocess.DEVNULL)
        lib = os.path.join(out, "libxeng_%s.so" % name)
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [ho], stderr=subprocess.DEVNULL)
        print("%-18s %s" % (name, why))
    with open(os.path.join(out, "variants.txt"), "w") as fh:
        for name in want:
            fh.write("%s\t%s\n" % (name, VARIANTS[name][0]))


if __name__ == "__main__":
    main()
