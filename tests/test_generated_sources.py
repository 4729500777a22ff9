"""Generated sources stay what their generators write: csrc/kernel_rowpair_rows.inc is committed text that
neither the Makefile nor any other test regenerates."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avisynth-jincresize_amd", "csrc")


def test_rowpair_rows_inc_is_what_its_generator_writes():
    out = subprocess.run([sys.executable, os.path.join(CSRC, "gen_rowpair_rows.py")], capture_output=True, text=True, check=True).stdout
    committed = open(os.path.join(CSRC, "kernel_rowpair_rows.inc"), encoding="utf-8").read()
    assert out == committed, "kernel_rowpair_rows.inc differs from gen_rowpair_rows.py's output: regenerate it (python3 gen_rowpair_rows.py > kernel_rowpair_rows.inc)"


def test_the_sdk_self_check_of_the_avisynth_shell_compiles_and_bites(tmp_path):
    """plugin/jincresize_avs.cpp checks struct layouts and enum values when it is NOT compiled against plugin/compat/avisynth_c.h
    (VERDICT r5 Next 9).  No SDK exists in the image, so the block is exercised with a copy of the compat header under another
    include guard (it then looks like 'some SDK header'): it must compile, and a changed enum value must stop the build."""
    compat = open(os.path.join(ROOT, "plugin", "compat", "avisynth_c.h"), encoding="utf-8").read()
    sdk = tmp_path / "sdk"
    sdk.mkdir()
    (sdk / "avisynth_c.h").write_text(compat.replace("JINCRESIZE_COMPAT_AVISYNTH_C_H", "SOME_SDK_AVISYNTH_C_H"))
    cmd = ["g++", "-std=c++17", "-fsyntax-only", f"-I{sdk}", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "plugin", "jincresize_avs.cpp")]
    ok = subprocess.run(cmd, capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-2000:]
    (sdk / "avisynth_c.h").write_text(compat.replace("JINCRESIZE_COMPAT_AVISYNTH_C_H", "SOME_SDK_AVISYNTH_C_H").replace("AVS_PLANAR_A = 1 << 4", "AVS_PLANAR_A = 1 << 8"))
    bad = subprocess.run(cmd, capture_output=True, text=True)
    assert bad.returncode != 0 and "AVS_PLANAR_*" in bad.stderr
