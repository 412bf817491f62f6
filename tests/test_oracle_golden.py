"""The oracle against every golden vector the reference's own tests hold for this path
(BwaMemIndexTest.java:38-127 -> tests/golden/reference_tests.json; fixture index files
src/test/resources/ref.fa.* -> tests/golden/rotavirus/)."""
import json
import os
import struct

import bwalib as B

GOLD = json.load(open(os.path.join(B.GOLDEN, "reference_tests.json")))


def _align(orc, img, reads, flag=0, pes=None):
    h = orc.open_index(img)
    opts = B.set_opt(orc.default_options(), flag=flag)
    buf = orc.align_raw(h, opts, B.pack_request(reads), pes)
    orc.destroy_index(h)
    return B.decode_response(buf, len(reads))


def test_opts_size_and_defaults(oracle):
    opts = oracle.default_options()
    assert len(opts) == GOLD["opts_size"] == 168          # BwaMemIndexTest.testOptsSize
    want = dict(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, pen_unpaired=17, pen_clip5=5, pen_clip3=5, w=100, zdrop=100,
                max_mem_intv=20, T=30, flag=0, min_seed_len=19, min_chain_weight=0, max_chain_extend=1 << 30, split_width=10,
                max_occ=500, max_chain_gap=10000, n_threads=1, chunk_size=10000000, mapQ_coef_fac=3, max_ins=10000,
                max_matesw=50, max_XA_hits=5, max_XA_hits_alt=200)
    for k, v in want.items():
        assert B.get_opt(opts, k) == v, k
    for k, v in dict(split_factor=1.5, mask_level=0.5, drop_ratio=0.5, XA_drop_ratio=0.8, mask_level_redun=0.95, mapQ_coef_len=50.0).items():
        assert abs(B.get_opt(opts, k) - v) < 1e-6, k
    mat = struct.unpack_from("25b", opts, 140)
    assert mat == (1, -4, -4, -4, -1, -4, 1, -4, -4, -1, -4, -4, 1, -4, -1, -4, -4, -4, 1, -1, -1, -1, -1, -1, -1)


def test_image_and_contig_names(oracle, rota_img):
    assert os.path.getsize(rota_img) == 3148               # SURVEY.md App. A.4 prediction ("(null)" -> "")
    h = oracle.open_index(rota_img)
    assert oracle.contig_names(h) == GOLD["contig_names"]
    oracle.destroy_index(h)


def test_single_end_known_answers(oracle, rota_img):       # testSimple, testMulti
    for case in GOLD["single_end"]:
        alns = _align(oracle, rota_img, case["reads"])
        for got, want in zip(alns, case["expect"]):
            assert len(got) == 1
            g = got[0]
            assert (g["pos"], g["ref_end"], g["seq_start"], g["seq_end"], g["cigar"], g["nm"], g["rid"], g["flag"]) == \
                   (want["refStart"], want["refEnd"], want["seqStart"], want["seqEnd"], want["cigar"], want["NM"], 0, want["flag"])


def test_paired_end_known_answers(oracle, rota_img):       # testPair x3
    pe = GOLD["paired_end"]
    for case in pe["cases"]:
        if case["peStats"] == "infer":
            pes = None
        elif case["peStats"] == "dont_infer":
            pes = B.pack_pestat(0, 0, 0, 0, failed=True)
        else:
            s = case["peStats"]
            pes = B.pack_pestat(s["low"], s["high"], float(s["average"]), float(s["std"]))
        alns = _align(oracle, rota_img, pe["reads"], flag=B.MEM_F_PE, pes=pes)
        for got, want, flag in zip(alns, pe["expect"], case["flags"]):
            assert len(got) == 1
            g = got[0]
            assert (g["pos"], g["ref_end"], g["seq_start"], g["seq_end"], g["cigar"], g["nm"], g["rid"], g["flag"]) == \
                   (want["refStart"], want["refEnd"], want["seqStart"], want["seqEnd"], want["cigar"], want["NM"], 0, flag)
            assert g["mpos"] == want["mateRefStart"] and g["tlen"] == want["tlen"]


def test_threads_do_not_change_results(oracle, rota_img):
    ref = open(os.path.join(B.GOLDEN, "rotavirus", "ref.fa")).read().split("\n", 1)[1].replace("\n", "").encode()
    reads = B.simulate_reads([("rotavirus", ref)], 200, length=70, seed=3, sub=0.03, indel=0.005)
    h = oracle.open_index(rota_img)
    req = B.pack_request(reads)
    one = oracle.align_raw(h, B.set_opt(oracle.default_options(), n_threads=1), req)
    four = oracle.align_raw(h, B.set_opt(oracle.default_options(), n_threads=4), req)
    oracle.destroy_index(h)
    assert one == four


def test_oracle_slice_entry_point_and_response_walker(oracle, rota_img):
    """checker utilities used by bench.py: oracle_createAlignmentsAt(read_id0 = 0) is oracle_createAlignments, and the C
    response walker finds the same record boundaries as the Python one"""
    import ctypes
    reads = [b"GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
             b"AATACTTCTTTTGAAGCTGCAGTTGTTGCTGCCTTCAACATTAGAATTAATGGGTATTCAATATGATT", b"ACGT" * 20, b"N" * 30, b""]
    h = oracle.open_index(rota_img)
    opts = oracle.default_options()
    req = B.pack_request(reads)
    want = oracle.align_raw(h, opts, req)
    fn = oracle.dll.oracle_createAlignmentsAt
    fn.restype = ctypes.c_void_p
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int64]
    rb = ctypes.create_string_buffer(req, len(req)); sz = ctypes.c_size_t()
    ob = ctypes.create_string_buffer(bytes(opts), B.OPT_SIZE)
    p = fn(h, ob, None, rb, ctypes.byref(sz), 0)
    assert ctypes.string_at(p, sz.value) == want
    oracle._free(p)
    offs = (ctypes.c_int64 * (len(reads) + 1))()
    ro = oracle.dll.oracle_response_offsets
    ro.restype = ctypes.c_int
    ro.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p]
    assert ro(want, len(want), len(reads), offs) == 0
    parts = B.split_response(want, len(reads))
    assert [offs[i + 1] - offs[i] for i in range(len(reads))] == [len(x) for x in parts]
    assert ro(want, len(want) - 1, len(reads), offs) == -1           # truncated buffer is reported
    oracle.destroy_index(h)


def test_alt_fixture_exercises_alt_paths(oracle, alt_genome):
    """the ALT fixture (tests/bwalib.py: synth_alt_genome) must actually reach the ALT-only rules: ALT hits as 0x800
    records after a primary-assembly record, ALT entries in XA, more XA entries than max_XA_hits when one is ALT, and
    a different response when the same sequences are indexed without the .alt file"""
    seqs, img, img0, alt_names, regions = alt_genome
    names = ["chr1_src", "chr2_src", "family", "chr1_alt1", "chr2_alt1", "chr1_alt2", "decoy"]
    reads = B.reads_from_regions(seqs, regions, names, 400, seed=3, sub=0.01, indel=0.001)
    n_alt = len(seqs) - len(alt_names)
    out = {}
    for image in (img, img0):
        ho = oracle.open_index(image)
        out[image] = B.decode_response(oracle.align_raw(ho, oracle.default_options(), B.pack_request(reads)), len(reads))
        oracle.destroy_index(ho)
    dec, dec0 = out[img], out[img0]
    assert sum(1 for r in dec for a in r if a["flag"] & 0x800 and a.get("rid", -1) >= n_alt) > 100
    assert sum(1 for r in dec0 for a in r if a["flag"] & 0x800) == 0
    assert sum(1 for r in dec for a in r if any(n in a.get("xa", "") for n in alt_names)) > 50
    assert sum(1 for r in dec for a in r if a.get("xa", "").count(";") > 5) > 0
    assert sum(r[0]["mapq"] for r in dec) > sum(r[0]["mapq"] for r in dec0)       # primary-assembly MAPQ no longer diluted by the ALT copy
