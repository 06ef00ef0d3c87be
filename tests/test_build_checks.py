"""The build's guard on the matrix-core search kernels (stereovision_amd/csrc/check_no_scratch.py): a kernel that spills a vector
register, or that has an instruction touching private memory, fails the build; a private segment that is reserved but never
touched does not.  Host logic only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "stereovision_amd", "csrc", "check_no_scratch.py")


def remarks(style, name, scratch, spill):
    lines = ["Function Name: %s" % name, "    VGPRs: 200", "    ScratchSize [bytes/lane]: %d" % scratch, "    VGPRs Spill: %d" % spill]
    if style == 0:      # plain compilation: file:line:col: remark: text [flag]
        return "".join("k.hip:7:1: remark: %s [-Rpass-analysis=kernel-resource-usage]\n" % t for t in lines)
    return "".join("remark: k.hip:7:0: %s [-Rpass-analysis=kernel-resource-usage]\n" % t for t in lines)   # with -save-temps


def run(tmp_path, text, asm=None, needle="mx_kernel"):
    r = tmp_path / "k.usage"
    r.write_text(text)
    cmd = [sys.executable, CHECK, str(r)]
    if asm is not None:
        a = tmp_path / "k.s"
        a.write_text(asm)
        cmd += ["--asm", str(a)]
    return subprocess.run(cmd + [needle], capture_output=True, text=True)


ASM = "other_kernel:\n\tscratch_store_dword off, v1, off\n\t.size other_kernel, 8\n_Z9mx_kernelv:\n\tv_mov_b32 v0, 0\n%s.LBB0_1:\n\ts_endpgm\n\t.size _Z9mx_kernelv, 16\n"


def test_clean_kernel_passes(tmp_path):
    for style in (0, 1):
        p = run(tmp_path, remarks(style, "_Z9mx_kernelv", 0, 0))
        assert p.returncode == 0 and "none spills" in p.stdout


def test_vector_spill_fails(tmp_path):
    for style in (0, 1):
        p = run(tmp_path, remarks(style, "_Z9mx_kernelv", 0, 3))
        assert p.returncode != 0 and "VGPRs Spill = 3" in p.stderr


def test_reserved_private_segment_needs_the_listing(tmp_path):
    text = remarks(1, "_Z9mx_kernelv", 36, 0)
    assert run(tmp_path, text).returncode != 0                                  # no listing: cannot tell
    p = run(tmp_path, text, ASM % "")
    assert p.returncode == 0 and "reserve a private segment" in p.stdout        # reserved, never touched (another kernel's access does not count)
    p = run(tmp_path, text, ASM % "\tscratch_load_dword v2, off, off offset:4\n")
    assert p.returncode != 0 and "1 private-memory instructions" in p.stderr


def test_other_kernels_are_ignored_and_a_missing_one_is_an_error(tmp_path):
    text = remarks(0, "_Z5otherv", 64, 9) + remarks(0, "_Z9mx_kernelv", 0, 0)
    assert run(tmp_path, text).returncode == 0
    assert run(tmp_path, remarks(0, "_Z5otherv", 0, 0)).returncode != 0


def test_missing_listing_is_an_error(tmp_path):
    r = tmp_path / "k.usage"
    r.write_text(remarks(0, "_Z9mx_kernelv", 0, 0))
    p = subprocess.run([sys.executable, CHECK, str(r), "--asm", str(tmp_path / "absent.s"), "mx_kernel"], capture_output=True, text=True)
    assert p.returncode != 0 and "cannot be read" in p.stderr


def test_buffer_access_through_the_private_descriptor_counts(tmp_path):
    text = remarks(1, "_Z9mx_kernelv", 36, 0)
    p = run(tmp_path, text, ASM % "\tbuffer_store_dword v2, off, s[0:3], 0 offset:4\n")
    assert p.returncode != 0 and "1 private-memory instructions" in p.stderr


# ---- check_mfma_hazards.py: distances between a matrix instruction (or a vector write) and the asm-block instructions that depend on it
HAZ = os.path.join(ROOT, "stereovision_amd", "csrc", "check_mfma_hazards.py")
MFMA = "\tv_mfma_f32_32x32x64_f8f6f4 v[%d:%d], v[50:53], v[60:63], v[%d:%d] cbsz:4 blgp:4\n"


def hazards(tmp_path, body, needle="mx_kernel"):
    a = tmp_path / "h.s"
    a.write_text("_Z5otherv:\n\tv_mfma_f32_32x32x64_f8f6f4 v[2:17], v[50:53], v[60:63], v[2:17] cbsz:4 blgp:4\n\tv_max_f32 v1, v2, v3\n\t.size _Z5otherv, 8\n"
                 "_Z9mx_kernelv:\n" + body + "\ts_endpgm\n\t.size _Z9mx_kernelv, 16\n")
    return subprocess.run([sys.executable, HAZ, str(a), needle], capture_output=True, text=True)


def test_reader_right_behind_the_product_fails(tmp_path):
    p = hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\t;;#ASMSTART\n\tv_max3_f32 v100, v2, v3, v4\n\t;;#ASMEND\n")
    assert p.returncode != 0 and "touches v2 11 wait states too early" in p.stderr
    # ten independent instructions in between are one too few, eleven are enough; s_nop k counts k + 1
    filler = "\tv_mov_b32 v120, v121\n"
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + filler * 10 + "\tv_max_f32 v100, v17, v101\n").returncode != 0
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + filler * 11 + "\tv_max_f32 v100, v17, v101\n").returncode == 0
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\ts_nop 15\n\ts_nop 7\n\tv_max_f32 v100, v17, v101\n").returncode == 0
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\ts_nop 9\n\tv_max_f32 v100, v17, v101\n").returncode != 0


def test_products_in_between_count_their_passes(tmp_path):
    """Back-to-back products are eight wait states apart (the pipe is busy): one in between is not enough for a reader of
    the first one's result (8 + 1 < 11), two are."""
    one = MFMA % (2, 17, 2, 17) + MFMA % (18, 33, 18, 33) + "\tv_max_f32 v100, v2, v101\n"
    two = MFMA % (2, 17, 2, 17) + MFMA % (18, 33, 18, 33) + MFMA % (34, 49, 34, 49) + "\tv_max_f32 v100, v2, v101\n"
    assert hazards(tmp_path, one).returncode != 0
    p = hazards(tmp_path, two)
    assert p.returncode == 0 and "3 matrix instructions" in p.stdout
    # overwriting a result register too early is caught like reading it; stores and LDS operations count as readers
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\tv_mov_b32 v5, v120\n").returncode != 0
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\tds_max_f32 v130, v9 offset:64\n").returncode != 0
    assert hazards(tmp_path, MFMA % (2, 17, 2, 17) + "\tds_read_b128 v[90:93], v130 offset:64\n").returncode == 0


def test_dpp_source_needs_two_wait_states(tmp_path):
    dpp = "\tv_max_f32_dpp v90, v80, v80 row_shl:4 row_mask:0xf bank_mask:0x5\n"
    assert hazards(tmp_path, "\tv_max_f32 v80, v1, v2\n" + dpp).returncode != 0
    assert hazards(tmp_path, "\tv_max_f32 v80, v1, v2\n\tv_mov_b32 v3, v4\n" + dpp).returncode != 0
    assert hazards(tmp_path, "\tv_max_f32 v80, v1, v2\n\ts_nop 1\n" + dpp).returncode == 0
    assert hazards(tmp_path, "\tv_max_f32 v80, v1, v2\n\tv_mov_b32 v3, v4\n\tv_mov_b32 v5, v4\n" + dpp).returncode == 0
    # the other operand of a DPP instruction is read in place: no distance needed
    assert hazards(tmp_path, "\tv_max_f32 v81, v1, v2\n\tv_max_f32_dpp v90, v80, v81 row_shl:4 row_mask:0xf bank_mask:0x5\n").returncode == 0
    # a vector write of EXEC needs five
    assert hazards(tmp_path, "\tv_cmpx_gt_u32_e64 s[2:3], v1, v2\n\ts_nop 3\n" + dpp).returncode != 0
    assert hazards(tmp_path, "\tv_cmpx_gt_u32_e64 s[2:3], v1, v2\n\ts_nop 4\n" + dpp).returncode == 0
    assert hazards(tmp_path, "\ts_or_b64 exec, exec, s[14:15]\n" + dpp).returncode == 0


def test_loop_back_edge_is_followed(tmp_path):
    """The last product of a loop body and the first reader of the next trip: only the path around the loop sees them together."""
    body = ("\tv_mov_b32 v120, v121\n.LBB0_1:\n\tv_max_f32 v100, v2, v101\n" + "\tv_mov_b32 v120, v121\n" * 12 + MFMA % (2, 17, 2, 17)
            + "\ts_cbranch_scc1 .LBB0_1\n" + "\tv_mov_b32 v120, v121\n" * 12)
    p = hazards(tmp_path, body)
    assert p.returncode != 0 and "touches v2" in p.stderr
    ok = body.replace("\ts_cbranch_scc1 .LBB0_1\n", "\ts_nop 15\n\ts_cbranch_scc1 .LBB0_1\n")
    assert hazards(tmp_path, ok).returncode == 0


def test_no_matching_kernel_is_an_error(tmp_path):
    assert hazards(tmp_path, "\tv_mov_b32 v1, v2\n", needle="absent_kernel").returncode != 0
