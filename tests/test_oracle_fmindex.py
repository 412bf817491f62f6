"""Pins the oracle's FM-index layer on the reference's fixture index by brute force: a sorted
suffix array of forward + reverse-complement ref.fa gives occ(), SA values and the SMEM set
from their definitions (SURVEY.md App. B items marked verified)."""
import ctypes
import os

import numpy as np

import bwalib as B


def _text():
    ref = open(os.path.join(B.GOLDEN, "rotavirus", "ref.fa")).read().split("\n", 1)[1].replace("\n", "")
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    fwd = [code[c] for c in ref]
    return fwd + [3 - c for c in reversed(fwd)]


def _suffix_array(t):
    n = len(t)
    return sorted(range(n + 1), key=lambda i: t[i:] + [-1] if i < n else [-1]) if False else \
        sorted(range(n + 1), key=lambda i: tuple(x + 1 for x in t[i:]))


def test_occ_and_sa_match_brute_force(oracle, rota_img):
    t = _text()
    n = len(t)
    sa = _suffix_array(t)                      # rank 0 = empty suffix (sentinel)
    assert sa[0] == n
    bwt = [t[p - 1] if p > 0 else -1 for p in sa]
    h = oracle.open_index(rota_img)
    occ4 = oracle.dll.o_bwt_occ4
    occ4.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    lookup = oracle.dll.o_bwt_sa
    lookup.restype = ctypes.c_uint64
    lookup.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    cnt = (ctypes.c_uint64 * 4)()
    run = [0, 0, 0, 0]
    for k in range(n + 1):
        if bwt[k] >= 0:
            run[bwt[k]] += 1
        occ4(h, k, cnt)
        assert list(cnt) == run, k
    for k in range(1, n + 1):
        assert lookup(h, k) == sa[k], k
    oracle.destroy_index(h)


def _brute_smems(t_str, q, min_len):
    """all super-maximal exact matches of q (codes; 4 = N) in the doubled text, with occurrence counts"""
    n = len(q)
    ends = []
    for i in range(n):
        e = i
        while e < n and q[e] < 4 and _occ(t_str, q[i:e + 1]) > 0:
            e += 1
        ends.append(e)
    out = []
    for i in range(n):
        if ends[i] > i and (i == 0 or ends[i] > ends[i - 1]) and ends[i] - i >= min_len:
            out.append((i, ends[i], _occ(t_str, q[i:ends[i]])))
    return out


def _occ(t_str, codes):
    s = "".join("ACGT"[c] for c in codes)
    cnt, pos = 0, t_str.find(s)
    while pos >= 0:
        cnt += 1
        pos = t_str.find(s, pos + 1)
    return cnt


def test_smem_pass1_equals_definition(oracle, rota_img):
    t = _text()
    t_str = "".join("ACGT"[c] for c in t)
    ref = [("rotavirus", t_str[: len(t) // 2].encode())]
    reads = B.simulate_reads(ref, 25, length=70, seed=8, sub=0.04, indel=0.005, n_rate=0.01)
    h = oracle.open_index(rota_img)
    collect = oracle.dll.oracle_collect_intv
    collect.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    # pass 1 only: split_width = 0 disables re-seeding, max_mem_intv = 0 disables the third pass
    opts = B.set_opt(oracle.default_options(), split_width=0, max_mem_intv=0, min_seed_len=12)
    ob = ctypes.create_string_buffer(bytes(opts), 168)
    out = (ctypes.c_uint64 * (4 * 256))()
    code = {65: 0, 67: 1, 71: 2, 84: 3, 78: 4}
    for rd in reads:
        q = [code[c] for c in rd]
        n = collect(h, ob, len(q), bytes(q), out, 256)
        got = [((out[4 * i + 3] >> 32), out[4 * i + 3] & 0xffffffff, out[4 * i + 2]) for i in range(n)]
        assert got == _brute_smems(t_str, q, 12)
    oracle.destroy_index(h)
