import os
import sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bwalib as B  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    B.build_oracle()
    return B.oracle_lib()


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("bwa"))


@pytest.fixture(scope="session")
def rota_img(oracle, workdir):
    img = os.path.join(workdir, "rota.img")
    assert oracle.create_index_file(os.path.join(B.GOLDEN, "rotavirus", "ref.fa"), img) == 0
    return img


@pytest.fixture(scope="session")
def hip_lib():
    """the product library; must exist in-tree (built by __graft_entry__.build())."""
    assert os.path.exists(B.HIP_LIB), "libbwamem_hip.so missing: run python -c 'import __graft_entry__ as g; g.build()'"
    return B.product_lib(emu=False)


def _build_genome(lib, workdir, name, total_bp, seqs=None, alt_names=None, **kw):
    import ctypes
    fa = os.path.join(workdir, name + ".fa")
    if seqs is None:
        seqs = B.synth_genome(total_bp, **kw)
    B.write_fasta(fa, seqs)
    if alt_names:
        B.write_alt_file(fa + ".alt", alt_names)
    build = lib.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    assert build(fa.encode(), fa.encode(), b"auto") == 0
    img = fa + ".img"
    assert lib.create_index_file(fa, img) == 0
    return seqs, img


@pytest.fixture(scope="session")
def small_genome(hip_lib, workdir):
    """300 kbp, 4 contigs, 15% diverged repeats, a few N runs (host index builder, no GPU needed)."""
    return _build_genome(hip_lib, workdir, "g300k", 300000, n_contigs=4, seed=7, repeat_frac=0.15, n_frac=0.001)


@pytest.fixture(scope="session")
def medium_genome(hip_lib, workdir):
    return _build_genome(hip_lib, workdir, "g3m", 3000000, n_contigs=6, seed=11, repeat_frac=0.08, n_frac=0.0005)


@pytest.fixture(scope="session")
def alt_genome(hip_lib, workdir):
    """primary assembly + ALT contigs listed in <prefix>.alt (and the same sequences without the .alt file, for contrast)
    -> (seqs, img, img_without_alt, alt_names, regions)"""
    seqs, alt_names, regions = B.synth_alt_genome()
    _, img = _build_genome(hip_lib, workdir, "galt", 0, seqs=seqs, alt_names=alt_names)
    _, img0 = _build_genome(hip_lib, workdir, "galt_noalt", 0, seqs=seqs)
    return seqs, img, img0, alt_names, regions


@pytest.fixture(scope="session")
def repeat_genome(hip_lib, workdir):
    """1.5 Mbp with a young 300-base family: 600 copies at 2 % divergence -> (seqs, img, copy starts)"""
    seqs, starts = B.synth_repeat_genome(n_copies=600, div=0.02)
    _, img = _build_genome(hip_lib, workdir, "grep", 0, seqs=seqs)
    return seqs, img, starts
