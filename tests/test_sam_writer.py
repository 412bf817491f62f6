"""SAM text on the native side (csrc/sam_writer.cpp): bwamem_hip_response_to_sam must write, for every record
of a response, the line an independent Python formatter builds from the decoded records (tests/bwalib.py: decode_response, i.e.
BwaMemAligner.java:215-307) under the rules stated in the writer's header.  CPU suite: the emulation build makes the responses."""
import ctypes
import re

import pytest

import bwalib as B


def _bind(lib):
    f = lib.dll.bwamem_hip_response_to_sam
    f.restype = ctypes.c_void_p
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
    h = lib.dll.bwamem_hip_sam_header
    h.restype = ctypes.c_void_p
    h.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    return f, h


def _to_sam(lib, h, req, resp, paired, names=None):
    f, _ = _bind(lib)
    sz = ctypes.c_size_t()
    arr = None
    if names is not None:
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    p = f(h, req, resp, len(resp), arr, 1 if paired else 0, ctypes.byref(sz))
    assert p, "the response did not parse against the request"
    out = ctypes.string_at(p, sz.value).decode()
    lib._free(p)
    return out


def _expected(reads, recs, contigs, paired, names=None):
    lines = []
    for r, alns in enumerate(recs):
        seq = reads[r].decode()
        name = names[r] if names else ("p%d" % (r >> 1) if paired else "r%d" % r)
        for k, a in enumerate(alns):
            flag = a["flag"]
            mapped = not flag & 4
            has_mate = flag & 9 == 1
            hard = k > 0 and mapped
            if mapped:
                rname, pos = contigs[a["rid"]], a["pos"] + 1
            elif has_mate and a["mrid"] >= 0:
                rname, pos = contigs[a["mrid"]], a["mpos"] + 1
            else:
                rname, pos = "*", 0
            cigar = a["cigar"] if mapped and a["cigar"] else "*"
            ops = re.findall(r"(\d+)([MIDNSHP=X])", cigar) if cigar != "*" else []
            c5 = int(ops[0][0]) if ops and ops[0][1] == "S" else 0
            c3 = int(ops[-1][0]) if len(ops) > 1 and ops[-1][1] == "S" else 0
            if hard:
                cigar = cigar.replace("S", "H")
            if has_mate and a["mrid"] >= 0:
                rnext, pnext = ("=" if (a["mrid"] == a.get("rid", -1) or not mapped) else contigs[a["mrid"]]), a["mpos"] + 1
            elif has_mate and mapped:
                rnext, pnext = "=", a["pos"] + 1
            else:
                rnext, pnext = "*", 0
            tlen = a["tlen"] if has_mate and mapped and a["mrid"] >= 0 else 0
            s = B.revcomp(seq.encode()).decode() if flag & 0x10 else seq
            if hard:
                s = s[c5:len(s) - c3]
            f = [name, str(flag), rname, str(pos), str(a["mapq"]), cigar, rnext, str(pnext), str(tlen), s if seq else "*", "*"]
            if mapped:
                f.append("NM:i:%d" % a["nm"])
                if a["md"]:
                    f.append("MD:Z:" + a["md"])
                f.append("AS:i:%d" % a["AS"])
                if a["XS"] >= 0:
                    f.append("XS:i:%d" % a["XS"])
                if a["xa"]:
                    f.append("XA:Z:" + a["xa"])
            lines.append("\t".join(f))
    return "\n".join(lines) + "\n"


def _check_lines(sam, reads_by_name):
    """every line on its own: 11 mandatory columns; SEQ as long as the CIGAR's query span (hard clips excluded); POS inside the contig"""
    for ln in sam.splitlines():
        c = ln.split("\t")
        assert len(c) >= 11
        if c[5] != "*":
            ops = re.findall(r"(\d+)([MIDNSHP=X])", c[5])
            assert "".join(n + o for n, o in ops) == c[5]
            assert sum(int(n) for n, o in ops if o in "MIS=X") == len(c[9])


def test_sam_writer_single_and_paired(oracle, small_genome):
    B.build_emu()
    emu = B.product_lib(emu=True)
    seqs, img = small_genome
    h = emu.open_index(img)
    contigs = emu.contig_names(h)
    # single-end: clean reads, chimeras (supplementary records with hard clips), reverse-strand reads, junk, an empty read
    reads = B.simulate_reads(seqs, 14, length=100, seed=3, sub=0.02, indel=0.004)
    g = seqs[0][1]
    reads += [g[3000:3060] + B.revcomp(g[9000:9070]), g[12000:12050] + g[20000:20080], b"ACGT" * 15, b""]
    req = B.pack_request(reads)
    opts = emu.default_options()
    resp = emu.align_raw(h, opts, req)
    recs = B.decode_response(resp, len(reads))
    assert any(len(a) > 1 for a in recs), "no read with several records: the clipping rule would go untested"
    sam = _to_sam(emu, h, req, resp, False)
    assert sam == _expected(reads, recs, contigs, False)
    _check_lines(sam, None)
    names = ["read_%d/x" % i for i in range(len(reads))]
    assert _to_sam(emu, h, req, resp, False, names) == _expected(reads, recs, contigs, False, names)
    # paired-end, including a pair with an unmappable mate and an odd trailing read
    pairs = B.simulate_pairs(seqs, 6, length=100, seed=5, ins_mean=300, ins_sd=30)
    pairs[3] = b"ACGT" * 25                                             # (nothing to rescue either: every 4-mer is everywhere)
    pairs.append(pairs[0])
    po = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
    preq = B.pack_request(pairs)
    presp = emu.align_raw(h, po, preq, B.pack_pestat(150, 450, 300.0, 30.0))
    precs = B.decode_response(presp, len(pairs) - 1) + [[]]
    psam = _to_sam(emu, h, preq, presp, True)
    assert psam == _expected(pairs, precs, contigs, True)
    _check_lines(psam, None)
    # a response that does not belong to the request is refused
    f, hdr = _bind(emu)
    sz = ctypes.c_size_t()
    assert not f(h, req, presp, len(presp), None, 0, ctypes.byref(sz))
    p = hdr(h, ctypes.byref(sz))
    text = ctypes.string_at(p, sz.value).decode()
    emu._free(p)
    assert text.startswith("@HD\tVN:1.6") and all(("@SQ\tSN:%s\tLN:%d" % (n, len(s))) in text for n, s in seqs)
    emu.destroy_index(h)
