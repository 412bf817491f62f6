"""Child process of tests/test_emu_parity.py::test_emu_sanitizers: loads the AddressSanitizer + UBSan flavour of the emulation
build (the unchanged product sources: host pipeline, index reader, every kernel's indexing) and runs single-end and
paired-end calls against the oracle.  Any sanitizer report aborts the process (UBSan is built with
-fno-sanitize-recover) and fails the test.  usage: sanitized_child.py <rotavirus.img> <small-genome.img> <small-genome.fa>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwalib as B  # noqa: E402

emu = B.Lib(os.path.join(B.ROOT, "tests", "emu", "_build", "libbwamem_emu_asan.so"), "jnibwa_")
orc = B.oracle_lib()
rota, small, fa = sys.argv[1:4]
seqs = []
for blk in open(fa).read().split(">")[1:]:
    name, _, body = blk.partition("\n")
    seqs.append((name.strip(), body.replace("\n", "").encode()))

golden = [b"GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
          b"GGCTTTTAATGCTTTTCAGTGCTAGGTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
          b"AATACTTCTTTTGAAGCTGCAGTTGTTGCTGCCTTCAACATTAGAATTAATGGGTATTCAATATGATT", b"ACGT" * 20, b"N" * 30, b"", b"A"]
cases = [(rota, golden, 0), (rota, golden[:6], B.MEM_F_PE),
         (small, B.simulate_reads(seqs, 4, length=150, seed=3, sub=0.03, indel=0.01, n_rate=0.01) + [b"", b"ACGTN" * 9], 0),
         (small, B.simulate_pairs(seqs, 2, length=100, seed=4, ins_mean=300, ins_sd=30), B.MEM_F_PE)]
for img, reads, flag in cases:
    h, ho = emu.open_index(img), orc.open_index(img)
    opts = B.set_opt(emu.default_options(), flag=flag)
    req = B.pack_request(reads)
    assert emu.align_raw(h, opts, req) == orc.align_raw(ho, opts, req), (img, flag)
    emu.destroy_index(h); orc.destroy_index(ho)
print("sanitized-ok")
