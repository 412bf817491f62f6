"""The device half of the index builder (csrc/k_index.hip: suffix array by MSD bucket sort on 32-base keys, BWT, occ checkpoints,
SA samples; reached through jnibwa_createReferenceIndex, which replaces bwa_idx_build at ...BwaMemIndex.c:42-63) must write
the very files the host builder writes -- and that one reproduces the reference's fixture files byte for byte."""
import ctypes
import os
import subprocess
import sys

import pytest

import bwalib as B

EXTS = ("amb", "ann", "bwt", "pac", "sa")


def _build(lib, fa, prefix, how, monkeypatch):
    monkeypatch.setenv("BWAMEM_HIP_INDEX_BUILDER", how)
    fn = lib.dll.jnibwa_createReferenceIndex
    fn.argtypes = [ctypes.c_char_p] * 3
    assert fn(fa.encode(), prefix.encode(), b"auto") == 0, how
    return {e: open(prefix + "." + e, "rb").read() for e in EXTS}


@pytest.mark.gpu
def test_device_builder_reproduces_the_reference_fixture(hip_lib, workdir, monkeypatch):
    src = os.path.join(B.GOLDEN, "rotavirus", "ref.fa")
    got = _build(hip_lib, src, os.path.join(workdir, "dev_rota"), "device", monkeypatch)
    for e in EXTS:
        assert got[e] == open(src + "." + e, "rb").read(), e           # src/test/resources/ref.fa.* of the reference


@pytest.mark.gpu
def test_device_builder_equals_host_builder(hip_lib, workdir, monkeypatch):
    """repeats (ties beyond the first 34 bases), a low-complexity tail (suffixes that run past the end of the text while still
    tied), several contigs, ambiguous bases"""
    seqs = B.synth_genome(900000, n_contigs=5, seed=6, repeat_frac=0.2, n_frac=0.001)
    seqs[-1] = (seqs[-1][0], seqs[-1][1] + b"C" * 90 + b"G" * 30 + b"A" * 200)
    seqs.append(("tandem", (b"ACGTTGCA" * 9 + b"T") * 300))
    fa = os.path.join(workdir, "devb.fa")
    B.write_fasta(fa, seqs)
    host = _build(hip_lib, fa, os.path.join(workdir, "devb_host"), "host", monkeypatch)
    dev = _build(hip_lib, fa, os.path.join(workdir, "devb_dev"), "device", monkeypatch)
    for e in EXTS:
        assert dev[e] == host[e], e


def test_device_builder_is_not_a_silent_fallback(hip_lib, workdir, monkeypatch):
    """without a device BWAMEM_HIP_INDEX_BUILDER=device fails instead of quietly building on the host (CPU suite)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is visible")
    monkeypatch.setenv("BWAMEM_HIP_INDEX_BUILDER", "device")
    fn = hip_lib.dll.jnibwa_createReferenceIndex
    fn.argtypes = [ctypes.c_char_p] * 3
    assert fn(os.path.join(B.GOLDEN, "rotavirus", "ref.fa").encode(), os.path.join(workdir, "nodev").encode(), b"auto") != 0


HL_CODE = r"""
import os, sys, time, ctypes
import torch
import numpy as np
sys.path[:0] = [%(tests)r, %(root)r, %(pkg)r]
import bwalib as B
import bench
import index_build_gpu as G
lib = B.product_lib()
dev = torch.device("cuda", 0)
codes, contigs = bench.synth_genome_humanlike(torch, dev, %(bp)d, 3, 0x5EED)
wd = %(workdir)r
G.write_image(os.path.join(wd, "hl_torch.img"), G.build_pieces(codes), contigs)
asc = np.frombuffer(b"ACGT", dtype=np.uint8)[codes.cpu().numpy()]
seqs, off = [], 0
for name, ln in contigs:
    seqs.append((name, asc[off:off + ln].tobytes())); off += ln
fa = os.path.join(wd, "hl_dev.fa")
B.write_fasta(fa, seqs)
del codes
torch.cuda.empty_cache()
os.environ["BWAMEM_HIP_INDEX_BUILDER"] = "device"
fn = lib.dll.jnibwa_createReferenceIndex
fn.argtypes = [ctypes.c_char_p] * 3
t = time.time()
assert fn(fa.encode(), fa.encode(), b"auto") == 0
print("device builder: %%.1f s for %%d bp" %% (time.time() - t, %(bp)d))
assert lib.create_index_file(fa, fa + ".img") == 0
a, b = open(fa + ".img", "rb").read(), open(os.path.join(wd, "hl_torch.img"), "rb").read()
assert len(a) == len(b) and a == b
print("device-index-ok")
"""


@pytest.mark.gpu
def test_device_builder_on_a_repeat_rich_16mbp_genome(workdir):
    """satellite arrays and young repeat families: tied groups that take dozens of 32-base rounds; the image must equal the one
    the (independent, torch) bench builder makes.  Own process: torch initialises HIP first."""
    code = HL_CODE % dict(tests=os.path.join(B.ROOT, "tests"), root=B.ROOT, pkg=B.PKG, workdir=workdir, bp=16_000_000)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "device-index-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
