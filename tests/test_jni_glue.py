"""csrc/jni_glue.c (the nine Java_org_broadinstitute_hellbender_utils_bwa_BwaMemIndex_* entry points + JNI_OnLoad, mirroring
src/main/c/org_broadinstitute_hellbender_utils_bwa_BwaMemIndex.c:43-165 and init.c:12-29) compiled against the stub jni.h
and driven through a fake JNIEnv: tests/jni_stub/jni_driver.c states what is checked.  CPU suite: linked against the
emulation build; GPU suite: against libbwamem_hip.so."""
import os
import subprocess

import pytest

import bwalib as B

STUB = os.path.join(B.ROOT, "tests", "jni_stub")


def _drive(flavour, workdir):
    if flavour == "emu":
        B.build_emu()
    subprocess.run(["make", "-s", "-C", STUB, flavour], check=True)
    r = subprocess.run([os.path.join(STUB, "_build", "jni_driver_" + flavour), os.path.join(B.GOLDEN, "rotavirus", "ref.fa"), str(workdir)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "jni-glue-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_jni_glue_compiles_and_runs_on_the_emulation(tmp_path):
    _drive("emu", tmp_path)


@pytest.mark.gpu
def test_jni_glue_on_the_hip_library(tmp_path):
    _drive("hip", tmp_path)


def test_jni_glue_exports_the_reference_symbols():
    """the symbols the unchanged Java class binds (BwaMemIndex.java native methods), as `nm` sees them in the compiled glue"""
    B.build_emu()
    subprocess.run(["make", "-s", "-C", STUB, "emu"], check=True)
    out = subprocess.run(["nm", "--defined-only", os.path.join(STUB, "_build", "jni_driver_emu")], capture_output=True, text=True, check=True).stdout
    for name in ("createReferenceIndex", "createIndexImageFile", "openIndex", "destroyIndex", "createDefaultOptions", "getRefContigNames",
                 "createAlignments", "destroyByteBuffer", "getVersion"):
        assert " T Java_org_broadinstitute_hellbender_utils_bwa_BwaMemIndex_" + name in out, name
    assert " T JNI_OnLoad" in out
