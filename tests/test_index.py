"""Index formats and the C-ABI surface (CPU only, no compute on a device)."""
import ctypes
import filecmp
import os
import re

import pytest

import bwalib as B


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(B.ROOT, "include", "bwamem_hip.h")).read()
    names = set(re.findall(r"\b((?:jnibwa|bwamem_hip)_[A-Za-z_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in sorted(names):
        assert hasattr(hip_lib.dll, n), n


def test_product_has_no_oracle_dependency():
    import subprocess
    out = subprocess.run(["ldd", B.HIP_LIB], capture_output=True, text=True).stdout
    assert "oracle" not in out and "emu" not in out
    syms = subprocess.run(["nm", "-D", B.HIP_LIB], capture_output=True, text=True).stdout
    assert "oracle_" not in syms


def test_builder_is_byte_exact_on_reference_fixture(hip_lib, workdir):
    """fasta -> 5 index files must equal src/test/resources/ref.fa.{amb,ann,bwt,pac,sa} byte for byte"""
    build = hip_lib.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    prefix = os.path.join(workdir, "kat.fa")
    assert build(os.path.join(B.GOLDEN, "rotavirus", "ref.fa").encode(), prefix.encode(), b"auto") == 0
    for ext in ("amb", "ann", "bwt", "pac", "sa"):
        assert filecmp.cmp(prefix + "." + ext, os.path.join(B.GOLDEN, "rotavirus", "ref.fa." + ext), shallow=False), ext
    assert build(b"/nonexistent.fa", prefix.encode(), b"auto") != 0
    assert build(os.path.join(B.GOLDEN, "rotavirus", "ref.fa").encode(), prefix.encode(), b"bogus") == -1


def test_image_writer_matches_oracle(hip_lib, oracle, workdir, rota_img):
    img = os.path.join(workdir, "rota_product.img")
    assert hip_lib.create_index_file(os.path.join(B.GOLDEN, "rotavirus", "ref.fa"), img) == 0
    assert open(img, "rb").read() == open(rota_img, "rb").read()
    assert hip_lib.create_index_file("/nonexistent/prefix", img) == 2         # jnibwa.c:131 -> 2


def test_two_contig_index_like_reference_test(hip_lib, oracle, workdir):
    """BwaMemIndexTest.testIndexReference: 45212 + 13415 bp contigs seq1, seq2 -> contig names (oracle reads our image)"""
    import numpy as np
    rng = np.random.default_rng(13)
    seqs = [("seq1", B.BASES[rng.integers(0, 4, 45212)].tobytes()), ("seq2", B.BASES[rng.integers(0, 4, 13415)].tobytes())]
    fa = os.path.join(workdir, "two.fasta")
    B.write_fasta(fa, seqs, width=60)
    build = hip_lib.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    assert build(fa.encode(), fa.encode(), b"auto") == 0
    img = fa + ".idx"
    assert hip_lib.create_index_file(fa, img) == 0
    h = oracle.open_index(img)
    assert oracle.contig_names(h) == ["seq1", "seq2"]
    reads = [seqs[0][1][1000:1150], B.revcomp(seqs[1][1][13000:13150])]
    alns = B.decode_response(oracle.align_raw(h, oracle.default_options(), B.pack_request(reads)), 2)
    assert (alns[0][0]["rid"], alns[0][0]["pos"], alns[0][0]["cigar"]) == (0, 1000, "150M")
    assert (alns[1][0]["rid"], alns[1][0]["pos"], alns[1][0]["cigar"], alns[1][0]["flag"]) == (1, 13000, "150M", 16)
    oracle.destroy_index(h)


def test_host_mirror_error_behaviour(hip_lib, workdir):
    import sys
    sys.path.insert(0, B.PKG)
    import bwamem
    import pytest
    with pytest.raises(bwamem.CouldNotReadImageException):
        bwamem.BwaMemIndex(os.path.join(workdir, "does-not-exist.img"))
    with pytest.raises(ValueError):
        bwamem.BwaMemIndex.createIndexImageFromIndexFiles(None, "x")
    with pytest.raises(ValueError):
        bwamem.BwaMemPairEndStats(0.5)
    s = bwamem.BwaMemPairEndStats(200, 10, 1, 600)
    assert (s.low, s.high, s.failed) == (1, 600, False) and bwamem.BwaMemPairEndStats.DO_NOT_INFER.failed
    assert "cb950614" in bwamem.BwaMemIndex.getBWAVersion()


def test_no_device_libm():
    """libm stays on the host: decisions that depend on log/erfc read host-built (glibc) tables, so no
    device object may contain the math library's code.  ocml is linked as bitcode and inlined, so there is no symbol to
    look for; its double-precision log/erfc/exp are recognisable by v_frexp_* (argument reduction), the single-precision
    ones by the transcendental unit's instructions."""
    import glob
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    objs = sorted(glob.glob(os.path.join(B.PKG, "_build", "k_*.o")))
    if not os.path.exists(objdump) or not objs:
        pytest.skip("no llvm-objdump / object files (the GPU box carries only the built library)")
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            c = os.path.join(tmp, os.path.basename(o))
            subprocess.run(["cp", o, c], check=True)
            subprocess.run([objdump, "--offloading", c], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
            code = glob.glob(c + ".*gfx950")
            assert code, "no gfx950 code object in " + o
            asm = subprocess.run([objdump, "-d", code[0]], check=True, capture_output=True, text=True).stdout
            assert len(asm) > 10000
            for ins in ("v_frexp_", "v_log_f", "v_exp_f", "v_sqrt_f", "v_rsq_f", "v_sin_f", "v_cos_f"):
                assert ins not in asm, "%s contains %s: a device math-library call" % (os.path.basename(o), ins)
