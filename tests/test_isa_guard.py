"""The built library's gfx950 code, disassembled: the mitigation of the open k_describe defect (DESIGN.md "Known defect") is a
compiler flag, so nothing but the generated ISA can say whether it still holds.  Fails if any packed f32 vector instruction
(v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_pk_mov_b32 on float pairs is not arithmetic and stays allowed) is in ANY kernel of
libmorb.so, and if the Makefile could drop the flags that results depend on when a caller sets HIPFLAGS."""
import os
import re
import struct
import subprocess
import pytest
import multi_orb_slam_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FORBIDDEN = re.compile(r"\bv_pk_(mul|fma|add)_f32\b")


def code_objects(path):
    """The gfx950 ELF images inside a HIP fat binary: every `__CLANG_OFFLOAD_BUNDLE__` block lists (offset, size, triple) entries."""
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        p = blob.find(magic, pos)
        if p < 0:
            return out
        q = p + len(magic)
        (n,) = struct.unpack_from("<Q", blob, q)
        q += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + tl].decode()
            q += tl
            if "amdgcn" in triple and size:
                assert "gfx950" in triple, triple
                out.append(blob[p + off:p + off + size])
        pos = p + 1


def disassemble(tmp_path):
    kernels = {}   # kernel symbol -> list of instruction lines
    for i, co in enumerate(code_objects(m.LIB_PATH)):
        f = tmp_path / ("co%d.elf" % i)
        f.write_bytes(co)
        txt = subprocess.run([OBJDUMP, "-d", str(f)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            mm = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if mm:
                cur = mm.group(1)
                kernels[cur] = []
            elif cur is not None and line.startswith("\t"):
                kernels[cur].append(line.split("//")[0].strip())
    return kernels


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_no_packed_f32_arithmetic_in_any_kernel(tmp_path):
    kernels = disassemble(tmp_path)
    names = " ".join(kernels)
    # the whole library was seen, not one object of it
    for must in ("k_describe", "k_fast_cells", "k_octree", "k_project", "k_hamming_matrix_mfma", "k_frame_build_small", "k_bow_transform"):
        assert must in names, must
    assert len(kernels) >= 55
    offenders = {k: [i for i in ins if FORBIDDEN.search(i)] for k, ins in kernels.items()}
    offenders = {k: v for k, v in offenders.items() if v}
    assert not offenders, "packed f32 arithmetic is back (was -fno-slp-vectorize dropped?): %s" % {k: v[:3] for k, v in offenders.items()}
    # Round 5 narrowed the defect to packed instructions whose LOW result takes the HIGH half of a source pair (op_sel:[..1..]) while
    # FP4 matrix instructions run on the part (profiles/r05/describe_defect.md).  The packed 16-bit integer instructions of
    # k_fast_cells only ever redirect the HIGH result of a scalar operand (op_sel_hi), which thousands of bit-exact rig runs cover;
    # a low-half redirect on ANY packed instruction is refused until somebody has shown that form to be safe.
    low_redirect = re.compile(r"\bv_pk_\w+\b.*\bop_sel:\[[01,]*1[01,]*\]")
    offenders = {k: [i for i in ins if low_redirect.search(i)] for k, ins in kernels.items()}
    offenders = {k: v for k, v in offenders.items() if v}
    assert not offenders, "a packed instruction redirects its low result (op_sel): %s" % {k: v[:3] for k, v in offenders.items()}
    # and the guard is able to see such an instruction at all: k_describe's rotation is there, as scalar-float multiplies
    desc = next(v for k, v in kernels.items() if "k_describe" in k)
    assert sum(i.startswith("v_mul_f32") for i in desc) >= 16
    assert sum(i.startswith("global_load_dwordx4") for i in desc) >= 4


def kernel_descriptors(co):
    """{kernel symbol: its 64-byte kernel descriptor} out of one gfx950 code object (ELF64 little endian): the `<kernel>.kd` symbols."""
    shoff, = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", co, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize) for i in range(shnum)]   # name, type, flags, addr, offset, size, link, info, align, entsize
    out = {}
    for name, typ, flags, addr, off, size, link, info, align, entsize in secs:
        if typ != 2 or not entsize:   # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for k in range(size // entsize):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", co, off + k * entsize)
            end = co.index(b"\0", stroff + st_name)
            sym = co[stroff + st_name:end].decode()
            if not sym.endswith(".kd") or st_shndx == 0 or st_shndx >= shnum:
                continue
            sec = secs[st_shndx]
            fo = sec[4] + (st_value - sec[3])
            out[sym[:-3]] = co[fo:fo + 64]
    return out


def test_fast_kernels_keep_f16_denormals():
    # k_fast_cells takes three-input extrema of 0..255 held in 16-bit lanes with v_pk_minimum3_f16 / v_pk_maximum3_f16: as f16 those are
    # denormals, and the instructions return them unchanged only while the kernel runs with f16 denormals preserved
    # (compute_pgm_rsrc1.FLOAT_DENORM_MODE_16_64 = 3: bits 19:18 of the dword at offset 48 of the kernel descriptor)
    seen = 0
    for co in code_objects(m.LIB_PATH):
        for name, kd in kernel_descriptors(co).items():
            if "k_fast_cells" not in name:
                continue
            rsrc1, = struct.unpack_from("<I", kd, 48)
            assert (rsrc1 >> 18) & 3 == 3, (name, hex(rsrc1))
            seen += 1
    assert seen >= 2   # <128, 8> and <256, 4>


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_fast_kernels_use_the_packed_three_input_extrema_and_matrix_kernels_no_accvgpr_moves(tmp_path):
    kernels = disassemble(tmp_path)
    fast = [v for k, v in kernels.items() if "k_fast_cells" in k]
    assert len(fast) >= 2
    for ins in fast:
        assert sum(i.startswith("v_pk_minimum3_f16") for i in ins) >= 30 and sum(i.startswith("v_pk_maximum3_f16") for i in ins) >= 30
        assert sum(i.startswith("v_mul_lo_u32") for i in ins) <= 6   # (index splits are 24-bit multiplies: csrc/extractor.hip)
    # the matrix instructions' accumulators live in ordinary vector registers (MORB_MFMA_IN_VGPRS): no v_accvgpr_read per key
    mfma = {k: v for k, v in kernels.items() if any(i.startswith("v_mfma") for i in v)}
    assert len(mfma) >= 7
    for k, ins in mfma.items():
        assert not any(i.startswith("v_accvgpr") for i in ins), k


def test_required_flags_survive_a_callers_hipflags():
    # `make -n HIPFLAGS=-O2` must still compile with the flags results depend on
    csrc = os.path.join(ROOT, "multi_orb_slam_amd", "csrc")
    out = subprocess.run(["make", "-n", "-B", "-C", csrc, "HIPFLAGS=-O2"], capture_output=True, text=True, check=True).stdout
    compiles = [l for l in out.splitlines() if " -c " in l]
    assert len(compiles) >= 9
    for l in compiles:
        for flag in ("-fno-slp-vectorize", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950"):
            assert flag in l, (flag, l)
