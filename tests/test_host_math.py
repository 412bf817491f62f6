"""Host-side checks of integer rewrites of upstream's floating-point expressions (compiled against the HIP emulation
shim with g++; no GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_div_plus_matches_double_expression(tmp_path):
    exe = str(tmp_path / "divplus_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(HERE, "emu"),
                           "-I" + os.path.join(ROOT, "gatk-bwamem-jni_amd", "csrc"), os.path.join(HERE, "emu", "divplus_check.cpp"),
                           "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "0 mismatches" in out.stdout
