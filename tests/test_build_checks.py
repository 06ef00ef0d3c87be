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
