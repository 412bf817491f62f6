"""The torch index builder (bench-scale path) must produce the same image, byte for byte, as the host
builder -- also when its chunked code paths (needed beyond 2^31 elements) are exercised."""
import ctypes
import os
import sys

import numpy as np
import torch

import bwalib as B

sys.path.insert(0, B.PKG)


def test_torch_builder_matches_host_builder(hip_lib, workdir, monkeypatch):
    import index_build_gpu as G
    monkeypatch.setattr(G, "CHUNK", 100003)                    # force many chunks
    seqs = B.synth_genome(260000, n_contigs=3, seed=5, repeat_frac=0.2)
    seqs[-1] = (seqs[-1][0], seqs[-1][1] + b"A" * 70 + b"T" * 40)      # low-complexity tail: stresses end-of-text ties
    fa = os.path.join(workdir, "tb.fa")
    B.write_fasta(fa, seqs)
    build = hip_lib.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    assert build(fa.encode(), fa.encode(), b"auto") == 0
    assert hip_lib.create_index_file(fa, fa + ".img") == 0
    lut = torch.zeros(256, dtype=torch.uint8)
    for k, v in {65: 0, 67: 1, 71: 2, 84: 3}.items():
        lut[k] = v
    fwd = lut[torch.from_numpy(np.frombuffer(b"".join(s for _, s in seqs), dtype=np.uint8).copy()).long()]
    pieces = G.build_pieces(fwd)
    G.write_image(fa + ".torch.img", pieces, [(n, len(s)) for n, s in seqs])
    assert open(fa + ".torch.img", "rb").read() == open(fa + ".img", "rb").read()


import pytest


@pytest.mark.gpu
def test_torch_builder_on_the_gpu_matches_host_builder(hip_lib, workdir, monkeypatch):
    """the same comparison with the builder running where the bench runs it (cuda:0), chunked paths forced"""
    import index_build_gpu as G
    monkeypatch.setattr(G, "CHUNK", 250007)
    seqs = B.synth_genome(900000, n_contigs=5, seed=6, repeat_frac=0.2)
    seqs[-1] = (seqs[-1][0], seqs[-1][1] + b"C" * 90 + b"G" * 30)
    fa = os.path.join(workdir, "tbg.fa")
    B.write_fasta(fa, seqs)
    build = hip_lib.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    assert build(fa.encode(), fa.encode(), b"auto") == 0
    assert hip_lib.create_index_file(fa, fa + ".img") == 0
    lut = torch.zeros(256, dtype=torch.uint8)
    for k, v in {65: 0, 67: 1, 71: 2, 84: 3}.items():
        lut[k] = v
    fwd = lut[torch.from_numpy(np.frombuffer(b"".join(s for _, s in seqs), dtype=np.uint8).copy()).long()].to("cuda:0")
    pieces = G.build_pieces(fwd)
    G.write_image(fa + ".torch.img", pieces, [(n, len(s)) for n, s in seqs])
    assert open(fa + ".torch.img", "rb").read() == open(fa + ".img", "rb").read()
