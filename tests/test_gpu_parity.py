"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
committed golden vectors.  Bit-exact is the bar: all arithmetic on the path is integer / byte /
index work (the few float/double decisions must round identically)."""
import json
import struct
import os
import sys

import pytest

import bwalib as B

pytestmark = pytest.mark.gpu
sys.path.insert(0, B.PKG)


@pytest.fixture(scope="module")
def bwamem():
    import bwamem
    return bwamem


@pytest.fixture(scope="module")
def rota_index(bwamem, workdir):
    img = os.path.join(workdir, "rota_hip.img")
    bwamem.BwaMemIndex.createIndexImageFromIndexFiles(os.path.join(B.GOLDEN, "rotavirus", "ref.fa"), img)
    index = bwamem.BwaMemIndex(img)
    yield index
    index.close()


def check(alignment, refStart, refEnd, seqStart, seqEnd, cigar, nMismatches, samFlag):
    # BwaMemIndexTest.testAlignment, :129-140
    assert alignment.getRefStart() == refStart
    assert alignment.getRefEnd() == refEnd
    assert alignment.getSeqStart() == seqStart
    assert alignment.getSeqEnd() == seqEnd
    assert alignment.getCigar() == cigar
    assert alignment.getNMismatches() == nMismatches
    assert alignment.getRefId() == 0
    assert alignment.getSamFlag() == samFlag


def test_opts_size(bwamem, rota_index):           # BwaMemIndexTest.testOptsSize
    aligner = bwamem.BwaMemAligner(rota_index)
    assert aligner.getOptsSize() == aligner.getExpectedOptsSize() == 168


def test_simple(bwamem, rota_index):              # BwaMemIndexTest.testSimple
    aligner = bwamem.BwaMemAligner(rota_index)
    alignments = aligner.alignSeqs(["GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT"])
    assert len(alignments) == 1 and len(alignments[0]) == 1
    check(alignments[0][0], 0, 70, 0, 70, "70M", 0, 0)


def test_multi(bwamem, rota_index):               # BwaMemIndexTest.testMulti
    aligner = bwamem.BwaMemAligner(rota_index)
    seqs = ["GGCTTTTAATGCTTTTCAGTGCTAGGTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
            "AATAATAGAGCTTACCATCTGCTGAGTAGACTCCATCTTGAGCAGCAACCACTGAAAAGCATTAAAAGCC",
            "AATACTTCTTTTGAAGCTGCAGTTGTTGCTGCCTTCAACATTAGAATTAATGGGTATTCAATATGATT"]
    alignments = aligner.alignSeqs(seqs)
    assert [len(a) for a in alignments] == [1, 1, 1]
    check(alignments[0][0], 0, 70, 0, 70, "70M", 3, 0)
    check(alignments[1][0], 0, 70, 0, 70, "70M", 0, 0x10)
    check(alignments[2][0], 70, 140, 0, 68, "32M2D36M", 2, 0)


def test_golden_vectors(bwamem, rota_index):
    """every SE vector of tests/golden/reference_tests.json (from BwaMemIndexTest.java:45-82)"""
    gold = json.load(open(os.path.join(B.GOLDEN, "reference_tests.json")))
    aligner = bwamem.BwaMemAligner(rota_index)
    for case in gold["single_end"]:
        alns = aligner.alignSeqs(case["reads"])
        for got, want in zip(alns, case["expect"]):
            assert len(got) == 1
            check(got[0], want["refStart"], want["refEnd"], want["seqStart"], want["seqEnd"], want["cigar"], want["NM"], want["flag"])


def _parity(hip, orc, img, reads, **optkw):
    h, ho = hip.open_index(img), orc.open_index(img)
    try:
        opts = B.set_opt(hip.default_options(), **optkw)
        req = B.pack_request(reads)
        got = hip.align_raw(h, opts, req)
        want = orc.align_raw(ho, opts, req)
        assert got is not None
        if got != want:
            sa, sb = B.split_response(got, len(reads)), B.split_response(want, len(reads))
            bad = [i for i in range(len(reads)) if sa[i] != sb[i]]
            msg = "%d/%d reads differ; first: read %d %r\n  hip    %r\n  oracle %r" % (
                len(bad), len(reads), bad[0], reads[bad[0]], B.decode_response(sa[bad[0]], 1), B.decode_response(sb[bad[0]], 1))
            pytest.fail(msg)
        B.check_against_stock(img, opts, req, got)          # LIBBWA_PATH / BWA_ORACLE_SRC: a stock libbwa as second checker
        return got
    finally:
        hip.destroy_index(h); orc.destroy_index(ho)


def test_parity_rotavirus(hip_lib, oracle, rota_img):
    seqs = [("rotavirus", open(os.path.join(B.GOLDEN, "rotavirus", "ref.fa")).read().split("\n", 1)[1].replace("\n", "").encode())]
    reads = B.simulate_reads(seqs, 400, length=70, seed=5, sub=0.03, indel=0.004)
    _parity(hip_lib, oracle, rota_img, reads)


def test_parity_small_genome(hip_lib, oracle, small_genome):
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 3000, length=150, seed=1, sub=0.02, indel=0.003, n_rate=0.002, random_frac=0.02)
    _parity(hip_lib, oracle, img, reads)


def test_parity_ragged_and_edge_cases(hip_lib, oracle, small_genome):
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 300, length=60, seed=2) + B.simulate_reads(seqs, 300, length=251, seed=3, sub=0.05, indel=0.01)
    reads += [b"", b"A", b"ACGT" * 4, b"N" * 150, b"ACGTN" * 30, b"acgtacgtacgtacgtacgtacgtacgt", seqs[0][1][:18], seqs[0][1][:19], seqs[0][1][100:120],
              seqs[1][1][-150:], B.revcomp(seqs[1][1][:150]), seqs[0][1][500:575] + seqs[2][1][900:975]]
    _parity(hip_lib, oracle, img, reads)


def test_empty_request(hip_lib, oracle, rota_img):
    h = hip_lib.open_index(rota_img)
    got = hip_lib.align_raw(h, hip_lib.default_options(), B.pack_request([]))
    hip_lib.destroy_index(h)
    assert got == b""


def test_parity_options(hip_lib, oracle, small_genome):
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 500, length=120, seed=9, sub=0.03, indel=0.005)
    _parity(hip_lib, oracle, img, reads, flag=B.MEM_F_ALL)
    _parity(hip_lib, oracle, img, reads, o_del=16, o_ins=16, b=9, pen_clip5=5, pen_clip3=5)     # setIntraCtgOptions
    _parity(hip_lib, oracle, img, reads, w=10, zdrop=20, T=40, min_seed_len=15, max_occ=20)
    _parity(hip_lib, oracle, img, reads, flag=B.MEM_F_NO_MULTI | B.MEM_F_PRIMARY5, XA_drop_ratio=0.5, max_XA_hits=2)


def test_multi_tile_equals_single_tile(hip_lib, oracle, small_genome, monkeypatch):
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 1000, length=100, seed=21)
    whole = _parity(hip_lib, oracle, img, reads)
    monkeypatch.setenv("BWAMEM_HIP_TILE", "97")
    tiled = _parity(hip_lib, oracle, img, reads)
    assert whole == tiled


def test_many_seeding_chunks_and_tiles(hip_lib, oracle, medium_genome, monkeypatch):
    """seeding runs in chunks of several tiles, one chunk ahead of the tile workers, into two interval stores that are
    reused every other chunk: many small chunks and tiles must give the single-launch bytes (and the oracle's)"""
    seqs, img = medium_genome
    reads = B.simulate_reads(seqs, 30000, length=150, seed=77, sub=0.02, indel=0.003, n_rate=0.002, random_frac=0.02)
    whole = _parity(hip_lib, oracle, img, reads)
    monkeypatch.setenv("BWAMEM_HIP_TILE", "1300")
    monkeypatch.setenv("BWAMEM_HIP_SEED_CHUNK", "3000")
    monkeypatch.setenv("BWAMEM_HIP_STREAMS", "4")
    assert _parity(hip_lib, oracle, img, reads) == whole
    monkeypatch.setenv("BWAMEM_HIP_SEED_AHEAD", "0")
    monkeypatch.setenv("BWAMEM_HIP_STREAMS", "2")
    assert _parity(hip_lib, oracle, img, reads) == whole


def test_parity_medium_genome(hip_lib, oracle, medium_genome):
    seqs, img = medium_genome
    reads = B.simulate_reads(seqs, 20000, length=150, seed=42)
    _parity(hip_lib, oracle, img, reads)


# ---------------------------------------------------------------- paired-end
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pair(bwamem, rota_index, mode):          # BwaMemIndexTest.testPair x3
    seqs = ["GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
            "TTGTTTTTAACACCAGAGTCATCCATCACATAATCAAATTTACTTTTAACTCTGGTAAATACTTCATTGT"]
    aligner = bwamem.BwaMemAligner(rota_index)
    aligner.alignPairs()
    if mode == 1:
        aligner.setProperPairEndStats(bwamem.BwaMemPairEndStats(200, 10, 1, 600))
    elif mode == 2:
        aligner.dontInferPairEndStats()
    else:
        aligner.inferPairEndStats()
    alignments = aligner.alignSeqs(seqs)
    assert [len(a) for a in alignments] == [1, 1]
    a0, a1 = alignments[0][0], alignments[1][0]
    check(a0, 0, 70, 0, 70, "70M", 0, 0x63 if mode == 1 else 0x61)
    assert a0.getMateRefStart() == 140 and a0.getTemplateLen() == 210
    check(a1, 140, 210, 0, 70, "70M", 0, 0x93 if mode == 1 else 0x91)
    assert a1.getMateRefStart() == 0 and a1.getTemplateLen() == -210


def _parity_pe(hip, orc, img, reads, pes=None, **optkw):
    h, ho = hip.open_index(img), orc.open_index(img)
    try:
        opts = B.set_opt(hip.default_options(), flag=B.MEM_F_PE | optkw.pop("flag", 0), **optkw)
        req = B.pack_request(reads)
        got = hip.align_raw(h, opts, req, pes)
        want = orc.align_raw(ho, opts, req, pes)
        assert got is not None
        if got != want:
            sa, sb = B.split_response(got, len(reads)), B.split_response(want, len(reads))
            bad = [i for i in range(len(reads)) if sa[i] != sb[i]]
            pytest.fail("%d/%d PE reads differ; first: read %d\n  hip    %r\n  oracle %r" % (
                len(bad), len(reads), bad[0], B.decode_response(sa[bad[0]], 1), B.decode_response(sb[bad[0]], 1)))
        B.check_against_stock(img, opts, req, got, pes)
        return got
    finally:
        hip.destroy_index(h); orc.destroy_index(ho)


def _damaged_pairs(seqs, n, seed, length=100, ins_mean=300):
    import random
    pairs = B.simulate_pairs(seqs, n, length=length, seed=seed, ins_mean=ins_mean, ins_sd=30, sub=0.01)
    rnd = random.Random(seed)
    for i in range(1, len(pairs), 6):                 # some mates too diverged to seed: mate rescue has to find them
        r = bytearray(pairs[i])
        for k in range(0, len(r), 7):
            r[k] = ord("ACGT"[rnd.randrange(4)])
        pairs[i] = bytes(r)
    for i in range(0, len(pairs), 50):                # some junk mates
        pairs[i] = bytes(ord("ACGT"[rnd.randrange(4)]) for _ in range(length))
    return pairs


def test_parity_pe_rescue_kernel_forms(hip_lib, oracle, small_genome):
    """mate rescue runs ksw_align2 wave-cooperatively: stripes in registers up to 10 segments (every 150 bp mate), in LDS
    beyond; 250 bp mates take the LDS form in 16-bit mode (score range >= 250), 180 bp mates in byte mode"""
    seqs, img = small_genome
    _parity_pe(hip_lib, oracle, img, _damaged_pairs(seqs, 300, 11, length=250, ins_mean=600))
    _parity_pe(hip_lib, oracle, img, _damaged_pairs(seqs, 300, 12, length=180, ins_mean=450))
    _parity_pe(hip_lib, oracle, img, _damaged_pairs(seqs, 300, 13, length=150, ins_mean=350))


def test_pe_call_in_two_steps_for_sharded_callers(hip_lib, oracle, small_genome):
    """bwamem_hip_batch_pe_begin / _candidates / bwamem_hip_pestat / _pe_finish (the exchange point of a call sharded over
    several GPUs): two shards on this GPU, statistics reduced over both -> the single-call response"""
    import ctypes, sys
    sys.path.insert(0, B.PKG)
    import sharding
    seqs, img = small_genome
    pairs = _damaged_pairs(seqs, 400, 21, ins_mean=280) + _damaged_pairs(seqs, 400, 22, ins_mean=420)
    opts = B.set_opt(hip_lib.default_options(), flag=B.MEM_F_PE)
    ho = oracle.open_index(img)
    want = oracle.align_raw(ho, opts, B.pack_request(pairs))
    oracle.destroy_index(ho)
    d = hip_lib.dll
    d.bwamem_hip_batch_upload.restype = ctypes.c_void_p
    d.bwamem_hip_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    d.bwamem_hip_batch_pe_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    d.bwamem_hip_batch_pe_candidates.restype = ctypes.c_size_t
    d.bwamem_hip_batch_pe_candidates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    d.bwamem_hip_pestat.restype = None
    d.bwamem_hip_pestat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    d.bwamem_hip_batch_pe_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    d.bwamem_hip_batch_result_bytes.restype = ctypes.c_size_t
    d.bwamem_hip_batch_result_bytes.argtypes = [ctypes.c_void_p]
    d.bwamem_hip_batch_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    d.bwamem_hip_batch_free.argtypes = [ctypes.c_void_p]
    h = hip_lib.open_index(img)
    ob = ctypes.create_string_buffer(bytes(opts), 168)
    shards, cand_d, cand_i = [], b"", b""
    try:
        for rank in range(2):
            lo, hi = sharding.shard_range(len(pairs), rank, 2, paired=True)
            req = B.pack_request(pairs[lo:hi])
            batch = d.bwamem_hip_batch_upload(h, req, len(req))
            assert batch and d.bwamem_hip_batch_pe_begin(h, ob, batch, lo) == 0
            n = d.bwamem_hip_batch_pe_candidates(batch, None, None)
            assert n == (hi - lo) // 2
            db, ib = ctypes.create_string_buffer(n), ctypes.create_string_buffer(8 * n)
            d.bwamem_hip_batch_pe_candidates(batch, db, ib)
            cand_d += db.raw[:n]; cand_i += ib.raw[:8 * n]
            shards.append(batch)
        pes = ctypes.create_string_buffer(128)
        d.bwamem_hip_pestat(ob, cand_d, cand_i, len(cand_d), pes)
        got = b""
        for batch in shards:
            assert d.bwamem_hip_batch_pe_finish(h, ob, pes, batch) == 0
            nb = d.bwamem_hip_batch_result_bytes(batch)
            out = ctypes.create_string_buffer(max(nb, 1))
            assert d.bwamem_hip_batch_download(batch, out) == 0
            got += out.raw[:nb]
        assert got == want
    finally:
        for batch in shards:
            d.bwamem_hip_batch_free(batch)
        hip_lib.destroy_index(h)


def test_parity_pe_inferred_stats(hip_lib, oracle, small_genome):
    seqs, img = small_genome
    _parity_pe(hip_lib, oracle, img, _damaged_pairs(seqs, 1500, 5))


def test_parity_pe_given_stats_and_flags(hip_lib, oracle, small_genome):
    seqs, img = small_genome
    pairs = _damaged_pairs(seqs, 600, 6)
    _parity_pe(hip_lib, oracle, img, pairs, pes=B.pack_pestat(150, 450, 300.0, 30.0))
    _parity_pe(hip_lib, oracle, img, pairs, pes=B.pack_pestat(0, 0, 0, 0, failed=True))
    _parity_pe(hip_lib, oracle, img, pairs, flag=B.MEM_F_NO_RESCUE)
    _parity_pe(hip_lib, oracle, img, pairs, flag=B.MEM_F_NOPAIRING)
    _parity_pe(hip_lib, oracle, img, pairs[:-1])       # odd read count: the unpaired tail produces no bytes


def test_pe_multi_tile(hip_lib, oracle, small_genome, monkeypatch):
    seqs, img = small_genome
    pairs = _damaged_pairs(seqs, 400, 7)
    whole = _parity_pe(hip_lib, oracle, img, pairs)
    monkeypatch.setenv("BWAMEM_HIP_TILE", "90")
    assert _parity_pe(hip_lib, oracle, img, pairs) == whole


def test_concurrent_calls_on_one_index(hip_lib, oracle, small_genome):
    """BwaMemIndex is shared by many Java threads, each with its own BwaMemAligner (BwaMemIndex.java:16-27): concurrent
    jnibwa_createAlignments calls on one index must each return their own correct response (the library serialises them per
    device)"""
    import threading
    seqs, img = small_genome
    batches = [B.simulate_reads(seqs, 400 + 50 * k, length=150, seed=100 + k, sub=0.02, indel=0.003) for k in range(6)]
    ho = oracle.open_index(img)
    want = [oracle.align_raw(ho, oracle.default_options(), B.pack_request(b)) for b in batches]
    oracle.destroy_index(ho)
    h = hip_lib.open_index(img)
    got = [None] * len(batches)
    def work(k):
        for _ in range(3):
            got[k] = hip_lib.align_raw(h, hip_lib.default_options(), B.pack_request(batches[k]))
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    hip_lib.destroy_index(h)
    assert got == want


def test_parity_long_reads(hip_lib, oracle, medium_genome, monkeypatch):
    """config-5 style reads: seed re-scoring (row a10), wide bands, long chains, big global alignments"""
    seqs, img = medium_genome
    reads = B.simulate_reads(seqs, 150, length=1000, seed=5, sub=0.06, indel=0.02, random_frac=0.0)
    reads += B.simulate_reads(seqs, 40, length=3000, seed=6, sub=0.08, indel=0.03, random_frac=0.0)
    reads += B.simulate_reads(seqs, 6, length=10000, seed=7, sub=0.08, indel=0.06, random_frac=0.0)
    got = _parity(hip_lib, oracle, img, reads)
    # the forms the packed 16-bit global alignments fall back to (32-bit diagonals in k_gcigar, rows in LDS behind mem_patch_reg)
    monkeypatch.setenv("BWAMEM_HIP_DEBUGK", "65536")
    assert _parity(hip_lib, oracle, img, reads) == got


def test_parity_reads_beyond_lds_rows(hip_lib, oracle, medium_genome, small_genome, monkeypatch):
    """beyond ~12 000 bases the DP rows of k_extend / k_gcigar live in global memory (k_extend<true>, k_gcigar<true>)"""
    seqs, img = medium_genome
    reads = B.simulate_reads(seqs, 3, length=16000, seed=8, sub=0.08, indel=0.06, random_frac=0.0)
    reads += B.simulate_reads(seqs, 2, length=40000, seed=9, sub=0.05, indel=0.04, random_frac=0.0)
    reads += B.simulate_reads(seqs, 20, length=150, seed=10)
    _parity(hip_lib, oracle, img, reads)
    seqs, img = small_genome                           # the same kernels forced on ordinary reads, many per workgroup
    monkeypatch.setenv("BWAMEM_HIP_DP_ROWS", "hbm")
    _parity(hip_lib, oracle, img, B.simulate_reads(seqs, 3000, length=150, seed=11, sub=0.03, indel=0.005))


# ---------------------------------------------------------------- ALT contigs (rows a9 / a14 / a17 / a19)
ALT_REGIONS = ["chr1_src", "chr2_src", "family", "chr1_alt1", "chr2_alt1", "chr1_alt2", "decoy"]


def test_parity_alt_contigs_single_end(hip_lib, oracle, alt_genome):
    """every production hg38 image carries ALT / decoy / HLA contigs: the is_alt chain-overlap rule, the second round of
    mem_mark_primary_se (ars_hash2 order, secondary = INT_MAX), XA has_alt / max_XA_hits_alt and ALT supplementary records"""
    seqs, img, img0, alt_names, regions = alt_genome
    reads = B.reads_from_regions(seqs, regions, ALT_REGIONS, 4000, seed=3, sub=0.01, indel=0.001)
    reads += B.reads_from_regions(seqs, regions, ALT_REGIONS, 1000, length=251, seed=4, sub=0.03, indel=0.004)
    got = _parity(hip_lib, oracle, img, reads)
    dec = B.decode_response(got, len(reads))
    n_alt = len(seqs) - len(alt_names)                  # contig ids >= n_alt are ALT
    assert sum(1 for r in dec for a in r if a["flag"] & 0x800 and a.get("rid", -1) >= n_alt) > 500     # ALT hits reported as supplementary
    assert sum(1 for r in dec for a in r if any(n in a.get("xa", "") for n in alt_names)) > 300         # ALT hits in XA
    assert sum(1 for r in dec for a in r if a.get("xa", "").count(";") > 5) > 5                         # has_alt lifts max_XA_hits
    assert sum(1 for r in dec if r[0].get("rid", -1) < n_alt and r[0]["mapq"] > 0 and len(r) > 1 and r[1].get("rid", -1) >= n_alt) > 300
    assert _parity(hip_lib, oracle, img0, reads) != got                                                  # the .alt file matters
    _parity(hip_lib, oracle, img, reads[:1500], flag=B.MEM_F_ALL)
    _parity(hip_lib, oracle, img, reads[:1500], max_XA_hits=1, max_XA_hits_alt=3)
    _parity(hip_lib, oracle, img, reads[:1500], max_XA_hits=2, max_XA_hits_alt=2, XA_drop_ratio=0.5, flag=B.MEM_F_NO_MULTI | B.MEM_F_PRIMARY5)


def test_parity_alt_contigs_paired_end(hip_lib, oracle, alt_genome):
    seqs, img, img0, alt_names, regions = alt_genome
    pairs = B.pairs_from_regions(seqs, regions, ALT_REGIONS[:6], 1500, length=100, seed=4, ins_mean=300, ins_sd=30)
    import random
    rnd = random.Random(5)
    for i in range(1, len(pairs), 6):                     # mates that need rescue next to ALT anchors
        r = bytearray(pairs[i])
        for k in range(0, len(r), 7):
            r[k] = ord("ACGT"[rnd.randrange(4)])
        pairs[i] = bytes(r)
    got = _parity_pe(hip_lib, oracle, img, pairs)
    dec = B.decode_response(got, len(pairs))
    n_alt = len(seqs) - len(alt_names)
    assert sum(1 for r in dec for a in r if a["flag"] & 0x800 and a.get("rid", -1) >= n_alt) > 200
    _parity_pe(hip_lib, oracle, img, pairs, pes=B.pack_pestat(150, 450, 300.0, 30.0))
    _parity_pe(hip_lib, oracle, img, pairs[:800], flag=B.MEM_F_NO_RESCUE)
    _parity_pe(hip_lib, oracle, img, pairs[:800], flag=B.MEM_F_ALL, max_XA_hits=1)


# ---------------------------------------------------------------- sharded calls and the bench launcher
def test_single_end_call_in_two_shards_with_read_id0(hip_lib, oracle, small_genome):
    """BASELINE.json config 4 in miniature: one logical single-end call aligned as two shards on the HIP library, each
    carrying the index of its first read (read_id0 = 0 / n/2).  Reads with tied alignment scores make the tie-break hash of
    mem_mark_primary_se (hash_64(id + i)) decide which copy is primary, so a shard that ignored read_id0 would differ."""
    import sharding
    seqs, img = small_genome
    g = seqs[0][1]
    reads = B.simulate_reads(seqs, 3000, length=120, seed=78, sub=0.01)
    reads += [g[5000:5120], g[9000:9100] + g[9000:9020]] * 600                 # exact repeats of the same read at many read indexes
    dup = bytearray(B.simulate_reads(seqs, 1, length=120, seed=79, sub=0.0, indel=0.0, n_rate=0.0, random_frac=0.0)[0])
    reads += [bytes(dup)] * 400
    ho = oracle.open_index(img)
    opts = oracle.default_options()
    want = oracle.align_raw(ho, opts, B.pack_request(reads))
    # the case can tell: the second half aligned as a call of its own (read_id0 = 0) gives other bytes for some read
    import ctypes
    fn = oracle.dll.oracle_createAlignmentsAt
    fn.restype = ctypes.c_void_p
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int64]
    half = len(reads) // 2
    def oracle_at(lo, hi, id0):
        req = B.pack_request(reads[lo:hi])
        rb = ctypes.create_string_buffer(req, len(req)); sz = ctypes.c_size_t()
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        return ctypes.string_at(fn(ho, ob, None, rb, ctypes.byref(sz), id0), sz.value)
    assert oracle_at(0, half, 0) + oracle_at(half, len(reads), half) == want
    tells = oracle_at(0, half, 0) + oracle_at(half, len(reads), 0) != want
    oracle.destroy_index(ho)
    h = hip_lib.open_index(img)
    try:
        got = b"".join(sharding.align_shard(hip_lib.dll, h, opts, reads, r, 2) for r in range(2))
        got3 = b"".join(sharding.align_shard(hip_lib.dll, h, opts, reads, r, 3) for r in range(3))
    finally:
        hip_lib.destroy_index(h)
    assert got == want and got3 == want
    assert tells, "fixture too weak: no read's records depend on its index within the call"


def test_bench_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no external launcher) must start two ranks itself and print an n_gpus = 2 line.
    BENCH_REHEARSAL puts both ranks on GPU 0 with gloo: the launcher, the shared image, the barriers and the gather of
    the per-rank times are what is exercised, not a measurement."""
    import subprocess
    env = dict(os.environ, BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(B.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "100000",
                        "--genome-bp", "8000000", "--contigs", "4", "--cpu-sample", "0", "--h2h-calls", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and len(out["per_rank"]["reads_per_s"]) == 2
    assert out["value"] > 0 and out["host_to_host"]["reads_per_s"] > 0 and out["host_to_host"]["identical_to_resident_response"] is True


def test_parity_long_reads_with_long_deletions(hip_lib, oracle, medium_genome, monkeypatch):
    """mem_patch_reg merges the two sides of a deletion of up to 2 w (4 w when the sides overlap), and bwa_gen_cigar2 then asks
    for a band as wide as the deletion.  k_gcigar's LDS row rings are sized for that; with the rings forced small the same jobs
    must come out of its rows-in-global-memory form (the safety net for bands wider than any ring)."""
    seqs, img = medium_genome
    g = max((s for _, s in seqs), key=len)
    reads = []
    for k, (length, gap) in enumerate([(20000, 190), (12000, 150), (4000, 150), (4000, 190)]):
        a = 5000 + 30000 * k
        cut = length // 2
        r = bytearray(g[a:a + cut] + g[a + cut + gap:a + gap + length])
        for p in range(97, len(r), 211):                              # a sprinkling of substitutions
            r[p] = ord("ACGT"[("ACGT".index(chr(r[p])) + 1) % 4]) if chr(r[p]) in "ACGT" else r[p]
        reads.append(bytes(r))
        reads.append(B.revcomp(bytes(r)))
    got = _parity(hip_lib, oracle, img, reads)
    dec = B.decode_response(got, len(reads))
    assert sum(1 for r in dec for a in r if "190D" in a.get("cigar", "") or "150D" in a.get("cigar", "")) >= 6, "regions were not merged across the deletions"
    monkeypatch.setenv("BWAMEM_HIP_GCIGAR_RING", "256")
    assert _parity(hip_lib, oracle, img, reads) == got


def test_parity_reads_in_a_repeat_family(hip_lib, oracle, repeat_genome):
    """reads inside a young 600-copy family: intervals of hundreds of occurrences (max_occ sampling, frac_rep), hundreds of chains
    per read through mem_chain_flt's quadratic filter (k_chain's packed kept list), many extensions and XA candidates"""
    seqs, img, starts = repeat_genome
    g = seqs[0][1]
    import random
    rnd = random.Random(3)
    reads = []
    for st in rnd.sample(starts, 300):
        off = rnd.randrange(-100, 250)                     # inside the copy, straddling its edge, or next to it
        r = bytearray(g[st + off:st + off + 150])
        for p in rnd.sample(range(150), rnd.randrange(0, 4)):
            r[p] = ord("ACGT"[("ACGT".index(chr(r[p])) + 1) % 4])
        reads.append(bytes(r) if rnd.random() < 0.5 else B.revcomp(bytes(r)))
    reads += B.simulate_reads(seqs, 200, length=150, seed=9)
    got = _parity(hip_lib, oracle, img, reads)
    dec = B.decode_response(got, len(reads))
    assert sum(1 for r in dec for a in r if a.get("xa")) > 50 and sum(1 for r in dec if r[0]["mapq"] < 30) > 50     # the family shows
    _parity(hip_lib, oracle, img, reads[:150], max_occ=50, flag=B.MEM_F_ALL)
    pairs = []
    for st in rnd.sample(starts, 100):                     # one mate in a copy, the other in unique sequence next to it
        a = st + rnd.randrange(-40, 120)
        pairs += [bytes(g[a:a + 100]), B.revcomp(bytes(g[a + 250:a + 350]))]
    _parity_pe(hip_lib, oracle, img, pairs)


def test_bench_on_an_existing_image_with_a_stock_library_hook(small_genome, tmp_path):
    """bench.py --image (BWAHIP_REF_IMG): an existing index image instead of the synthetic genome, reads sampled from its packed
    reference; LIBBWA_PATH: the CPU baseline / checker taken from a library with the reference's jnibwa_* ABI.  No stock libbwa or
    GATK image exists offline, so the hooks are pointed at this repo's own image and library: what is exercised is the plumbing."""
    import subprocess
    seqs, img = small_genome
    env = dict(os.environ, LIBBWA_PATH=B.HIP_LIB)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_REHEARSAL"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(B.ROOT, "bench.py"), "--image", img, "--reads", "20000", "--steps", "1", "--warmup", "0",
                        "--cpu-sample", "3000", "--cpu-reps", "1", "--h2h-calls", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["genome"] == "image" and out["config"]["genome_bp"] == sum(len(s) for _, s in seqs)
    assert out["cpu_baseline"]["kind"] == "reference" and out["parity_sample"]["frac_identical_records"] == 1.0
    assert out["host_to_host"]["identical_to_resident_response"] is True


def test_paired_end_on_repeat_rich_genome(tmp_path):
    """Pairs drawn from a genome with human-like repeat content (bench.py --genome humanlike, scaled down): reads carry many
    regions, pairs ask for far more mate-rescue alignments than a tile's first guess holds, so the rescue job list is resized
    (the path that once ran half-written lists: an illegal memory access on the device), and the records must still be the
    oracle's.  Two calls through jnibwa_createAlignments as well: sizes learned from the first one."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(B.ROOT, "bench.py"), "--paired", "--genome", "humanlike", "--genome-bp", "16000000", "--contigs", "3",
                        "--reads", "60000", "--steps", "1", "--warmup", "1", "--cpu-sample", "60000", "--cpu-reps", "1", "--h2h-calls", "2"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["parity_sample"]["reads"] == 60000 and out["parity_sample"]["frac_identical_records"] == 1.0, out["parity_sample"]
    assert out["host_to_host"]["identical_to_resident_response"] is True


HUMANLIKE_CODE = r"""
import os, struct, sys
import torch                                     # (first: the process then has one HIP runtime, torch's, as in bench.py)
sys.path[:0] = [%(tests)r, %(root)r, %(pkg)r]
import bwalib as B
import bench
import index_build_gpu as G
hip_lib, oracle = B.product_lib(), B.oracle_lib()
dev = torch.device("cuda", 0)
codes, contigs = bench.synth_genome_humanlike(torch, dev, 16_000_000, 3, 0x5EED)
img = os.path.join(%(workdir)r, "hl16.img")
G.write_image(img, G.build_pieces(codes), contigs)
n_se, n_pe = 100_000, 60_000
se = bench.synth_reads(torch, dev, codes, contigs, n_se, 150, 42)
pe = bench.synth_pairs(torch, dev, codes, contigs, n_pe // 2, 150, 43)
del codes
torch.cuda.empty_cache()
h, ho = hip_lib.open_index(img), oracle.open_index(img)
assert h and ho
cores = min(16, len(os.sched_getaffinity(0)))
for payload, n, flag, stats in ((se, n_se, 0, [None]), (pe, n_pe, B.MEM_F_PE, [None, B.pack_pestat(200, 600, 400.0, 50.0)])):
    req = struct.pack("<i", n) + payload.cpu().numpy().tobytes()
    for pes in stats:
        opts = B.set_opt(hip_lib.default_options(), flag=flag)
        want = oracle.align_raw(ho, B.set_opt(bytearray(opts), n_threads=cores), req, pes)
        got = hip_lib.align_raw(h, opts, req, pes)
        assert got is not None
        if got != want:
            a, b = B.split_response(got, n), B.split_response(want, n)
            bad = [i for i in range(n) if a[i] != b[i]]
            raise AssertionError("%%d of %%d reads differ (first: %%s), flag %%d, statistics %%s" %% (len(bad), n, bad[:10], flag, "supplied" if pes else "inferred"))
        recs = B.decode_response(got, n)
        # the workload is what it claims to be: mapped reads with MAPQ 0 (several equally good places) occur by the hundred
        assert sum(1 for r in recs if r and r[0].get("mapq", 60) == 0 and not r[0]["flag"] & 4) > 50
hip_lib.destroy_index(h); oracle.destroy_index(ho)
print("humanlike-parity-ok")
"""


@pytest.mark.gpu
def test_parity_humanlike_16mbp_single_and_paired_end(hip_lib, oracle, workdir):
    """A 16 Mbp genome with human-like repeat content (SINE-/LINE-like families, satellite arrays, microsatellites, segmental
    duplications: bench.py's generator, scaled down), index built on the device, 100 000 single-end reads and 60 000 paired-end
    reads (statistics inferred, then supplied) straight through the C ABI against the oracle, every record compared.  The small
    genomes of the other tests have nothing like a satellite array: reads with hundreds of regions, pairs with dozens of
    rescue anchors and the wavefront forms of chain filtering and mate rescue are only reached here.  Own process: the genome
    and the index are made with torch on the GPU, and torch has to initialise HIP before this library does."""
    import subprocess
    code = HUMANLIKE_CODE % dict(tests=os.path.join(B.ROOT, "tests"), root=B.ROOT, pkg=B.PKG, workdir=workdir)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "humanlike-parity-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
