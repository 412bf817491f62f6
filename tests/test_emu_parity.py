"""CPU-side check of the kernels' logic: the UNCHANGED product sources compiled against the
host emulation of the HIP execution model in tests/emu/ (test infrastructure only; never
shipped, never a fallback) must reproduce the oracle byte for byte.  Kept small: the fiber
emulation of wavefront collectives is slow.  The real parity gate is tests/test_gpu_parity.py."""
import os

import pytest

import bwalib as B


@pytest.fixture(scope="module")
def emu():
    B.build_emu()
    return B.product_lib(emu=True)


def _cmp(emu, oracle, img, reads, **optkw):
    h, ho = emu.open_index(img), oracle.open_index(img)
    opts = B.set_opt(emu.default_options(), **optkw)
    req = B.pack_request(reads)
    got, want = emu.align_raw(h, opts, req), oracle.align_raw(ho, opts, req)
    emu.destroy_index(h); oracle.destroy_index(ho)
    assert got is not None
    assert got == want
    B.check_against_stock(img, opts, req, want)         # LIBBWA_PATH / BWA_ORACLE_SRC: the oracle itself against a stock libbwa


def test_emu_golden_reads(emu, oracle, rota_img):
    reads = [b"GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
             b"GGCTTTTAATGCTTTTCAGTGCTAGGTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
             b"AATAATAGAGCTTACCATCTGCTGAGTAGACTCCATCTTGAGCAGCAACCACTGAAAAGCATTAAAAGCC",
             b"AATACTTCTTTTGAAGCTGCAGTTGTTGCTGCCTTCAACATTAGAATTAATGGGTATTCAATATGATT", b"ACGT" * 20, b"N" * 30, b""]
    _cmp(emu, oracle, rota_img, reads)


def test_emu_small_genome(emu, oracle, small_genome):
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 24, length=150, seed=1, sub=0.02, indel=0.004, n_rate=0.002, random_frac=0.05)
    reads += B.simulate_reads(seqs, 8, length=251, seed=3, sub=0.05, indel=0.01)
    _cmp(emu, oracle, img, reads)
    _cmp(emu, oracle, img, reads[:12], flag=B.MEM_F_ALL, w=10, T=20)


def test_emu_paired_end(emu, oracle, rota_img, small_genome):
    pr = [b"GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT",
          b"TTGTTTTTAACACCAGAGTCATCCATCACATAATCAAATTTACTTTTAACTCTGGTAAATACTTCATTGT"]
    seqs, img = small_genome
    pairs = B.simulate_pairs(seqs, 12, length=100, seed=5, ins_mean=300, ins_sd=30)
    pairs[3] = bytes(c if i % 7 else 65 for i, c in enumerate(pairs[3]))      # a mate that needs rescue
    for image, reads, stats in [(rota_img, pr, [None, B.pack_pestat(1, 600, 200.0, 10.0), B.pack_pestat(0, 0, 0, 0, failed=True)]),
                                (img, pairs, [None, B.pack_pestat(150, 450, 300.0, 30.0)])]:
        h, ho = emu.open_index(image), oracle.open_index(image)
        for pes in stats:
            opts = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
            req = B.pack_request(reads)
            assert emu.align_raw(h, opts, req, pes) == oracle.align_raw(ho, opts, req, pes)
        emu.destroy_index(h); oracle.destroy_index(ho)


def test_emu_concurrent_tiles(emu, oracle, small_genome, monkeypatch):
    """several tiles in flight (one host thread + stream each) must give the single-tile bytes"""
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 30, length=100, seed=9, sub=0.02, indel=0.003)
    monkeypatch.setenv("BWAMEM_HIP_TILE", "7")
    monkeypatch.setenv("BWAMEM_HIP_STREAMS", "3")
    _cmp(emu, oracle, img, reads)
    monkeypatch.setenv("BWAMEM_HIP_SEED_CHUNK", "10")               # seeding chunks of one tile: both interval stores get reused
    _cmp(emu, oracle, img, reads)


def test_emu_long_reads_seed_rescoring(emu, oracle, small_genome, monkeypatch):
    """reads long enough (5.5 ln L <= 0.05 L) to go through mem_flt_chained_seeds / mem_seed_sw (row a10)"""
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 2, length=800, seed=5, sub=0.06, indel=0.02)
    _cmp(emu, oracle, img, reads)
    # beyond 1000 bases region post-processing runs one wavefront per read (k_post1<true>): a 45-base deletion makes it merge
    # two regions through the wave-parallel patch alignment
    g = seqs[0][1]
    rd = bytearray(g[20000:20620] + g[20665:21300])
    for p in range(50, len(rd), 173):
        rd[p] = ord("ACGT"[("ACGT".index(chr(rd[p])) + 1) % 4]) if chr(rd[p]) in "ACGT" else rd[p]
    _cmp(emu, oracle, img, [bytes(rd), B.revcomp(bytes(rd))])
    # the same through the forms the packed 16-bit global alignments fall back to (32-bit diagonals in k_gcigar, rows in LDS behind
    # mem_patch_reg): what a job takes whose values do not fit, or options whose penalties leave no room
    monkeypatch.setenv("BWAMEM_HIP_DEBUGK", "65536")
    _cmp(emu, oracle, img, [bytes(rd)])


def test_emu_seed_work_queue_and_spill(emu, oracle, small_genome, monkeypatch):
    """k_seed: one resident wave pulling 150 reads from the tile queue (lanes refill as they finish), candidate stacks
    of 3 LDS entries so nearly every search spills to the global area"""
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 130, length=70, seed=21, sub=0.03, indel=0.004, n_rate=0.01, random_frac=0.1)
    reads += [b"ACGT" * 20, b"N" * 30, b"", b"A" * 18, b"ACGTN" * 20] * 4
    # chimeras of six 40-base pieces: more long unique matches than the LDS list of pass-2 candidates holds
    g = seqs[0][1] if isinstance(seqs[0], tuple) else seqs[0]
    for k in range(3):
        reads.append(b"".join(bytes(g[1000 * (7 * j + k + 1):1000 * (7 * j + k + 1) + 40]) for j in range(6)))
    monkeypatch.setenv("BWAMEM_HIP_SEED_WPC", "1")
    monkeypatch.setenv("BWAMEM_HIP_SEED_K", "3")
    _cmp(emu, oracle, img, reads)


def test_emu_mate_rescue_kernel_forms(emu, oracle, small_genome):
    """wave-cooperative ksw_align2 of mate rescue: register stripes (<= 10 segments) and LDS stripes (longer mates, 16-bit mode)"""
    seqs, img = small_genome
    for length, ins in ((250, 600), (150, 350)):
        pairs = B.simulate_pairs(seqs, 8, length=length, seed=40 + length, ins_mean=ins, ins_sd=30)
        for k in (1, 6, 11):                                          # mates that do not seed: every 7th base replaced
            pairs[k] = bytes(c if i % 7 else 65 for i, c in enumerate(pairs[k]))
        h, ho = emu.open_index(img), oracle.open_index(img)
        opts = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
        req = B.pack_request(pairs)
        pes = B.pack_pestat(ins - 150, ins + 150, float(ins), 30.0)
        assert emu.align_raw(h, opts, req, pes) == oracle.align_raw(ho, opts, req, pes)
        emu.destroy_index(h); oracle.destroy_index(ho)


def test_emu_dp_rows_in_global_memory(emu, oracle, rota_img, small_genome, monkeypatch):
    """reads too long for LDS rows (beyond ~12 000 bases) run k_extend / k_gcigar with their rows in global memory; the same
    path forced on ordinary reads must give the same bytes, and a 20 kb read must go through"""
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 12, length=150, seed=9, sub=0.03, indel=0.01) + B.simulate_reads(seqs, 4, length=400, seed=10, sub=0.05, indel=0.02)
    monkeypatch.setenv("BWAMEM_HIP_DP_ROWS", "hbm")
    _cmp(emu, oracle, img, reads)
    monkeypatch.delenv("BWAMEM_HIP_DP_ROWS")
    _cmp(emu, oracle, rota_img, [b"ACGT" * 5000, b"ACGT" * 20])


ALT_REGIONS = ["chr1_src", "chr2_src", "family", "chr1_alt1", "chr2_alt1", "chr1_alt2", "decoy"]


def test_emu_alt_contigs(emu, oracle, alt_genome):
    """ALT-aware paths (rows a9/a14/a17/a19): is_alt chains, the second round of mem_mark_primary_se, XA has_alt /
    max_XA_hits_alt, ALT supplementary records, single- and paired-end"""
    seqs, img, img0, alt_names, regions = alt_genome
    reads = B.reads_from_regions(seqs, regions, ALT_REGIONS, 40, seed=3, sub=0.01, indel=0.001)
    _cmp(emu, oracle, img, reads)
    _cmp(emu, oracle, img, reads[:16], flag=B.MEM_F_ALL)
    _cmp(emu, oracle, img, reads[:16], max_XA_hits=1, max_XA_hits_alt=3)
    pairs = B.pairs_from_regions(seqs, regions, ALT_REGIONS[:6], 10, length=100, seed=4, ins_mean=300, ins_sd=30)
    h, ho = emu.open_index(img), oracle.open_index(img)
    opts = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
    for pes in (None, B.pack_pestat(150, 450, 300.0, 30.0)):
        assert emu.align_raw(h, opts, B.pack_request(pairs), pes) == oracle.align_raw(ho, opts, B.pack_request(pairs), pes)
    emu.destroy_index(h); oracle.destroy_index(ho)


def test_emu_streamed_request_stretches(emu, oracle, small_genome, monkeypatch):
    """jnibwa_createAlignments streams the request to the device in stretches cut by read count and by bytes (whole reads,
    whole pairs); the read offsets come from the device NUL scan.  Tiny stretches must give the same bytes."""
    seqs, img = small_genome
    reads = B.simulate_reads(seqs, 40, length=100, seed=33, sub=0.02, indel=0.003) + [b"", b"A", b"", b"ACGT" * 30, b""]
    monkeypatch.setenv("BWAMEM_HIP_UPLOAD_BYTES", "700")
    _cmp(emu, oracle, img, reads)
    pairs = B.simulate_pairs(seqs, 9, length=100, seed=34, ins_mean=300, ins_sd=30)
    monkeypatch.setenv("BWAMEM_HIP_UPLOAD_BYTES", "301")           # three reads' worth: stretches must still end on a pair
    h, ho = emu.open_index(img), oracle.open_index(img)
    opts = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
    for rd in (pairs, pairs[:-1]):
        assert emu.align_raw(h, opts, B.pack_request(rd)) == oracle.align_raw(ho, opts, B.pack_request(rd))
    emu.destroy_index(h); oracle.destroy_index(ho)


def test_stock_libbwa_hook(emu, oracle, rota_img, monkeypatch):
    """LIBBWA_PATH makes a library with the reference's jnibwa_* ABI a second checker (tests/bwalib.py: stock_libbwa).
    No stock libbwa exists in this offline image, so the hook is exercised by pointing it at a library that has the ABI --
    the emulation build -- and must be a clean no-op when unset."""
    reads = [b"GGCTTTTAATGCTTTTCAGTGGTTGCTGCTCAAGATGGAGTCTACTCAGCAGATGGTAAGCTCTATTATT", b"ACGT" * 20]
    req = B.pack_request(reads)
    ho = oracle.open_index(rota_img)
    want = oracle.align_raw(ho, oracle.default_options(), req)
    oracle.destroy_index(ho)
    monkeypatch.delenv("LIBBWA_PATH", raising=False)
    monkeypatch.delenv("BWA_ORACLE_SRC", raising=False)
    assert B.stock_libbwa() is None and B.check_against_stock(rota_img, oracle.default_options(), req, want) is False
    monkeypatch.setenv("LIBBWA_PATH", B.EMU_LIB)
    assert B.check_against_stock(rota_img, oracle.default_options(), req, want) is True
    with pytest.raises(AssertionError):
        B.check_against_stock(rota_img, oracle.default_options(), req, want + b"x")
    monkeypatch.setenv("LIBBWA_PATH", "/nonexistent/libbwa.Linux.so")
    with pytest.raises(FileNotFoundError):
        B.stock_libbwa()


def test_emu_sanitizers(oracle, rota_img, small_genome):
    """AddressSanitizer + UBSan over the product's host code and kernel indexing (tests/emu `make asan`), in a child process
    that preloads the sanitizer runtimes"""
    import subprocess
    import sys
    B.make(os.path.join(B.ROOT, "tests", "emu"), "asan")
    libs = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(x) and os.path.exists(x) for x in libs):
        pytest.skip("no sanitizer runtimes next to this gcc")
    seqs, img = small_genome
    env = dict(os.environ, LD_PRELOAD=":".join(libs), ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, os.path.join(B.ROOT, "tests", "emu", "sanitized_child.py"), rota_img, img, img[:-4]],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitized-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_emu_repeat_family(emu, oracle, workdir, monkeypatch):
    """reads in a 150-copy family: more chains per read than a wavefront has lanes, all kept by mem_chain_flt -- its overlap
    loop run by the wavefront (k_chain.hip: chain_flt_wave) and, second pass, by the read's lane"""
    import ctypes
    seqs, starts = B.synth_repeat_genome(total_bp=150000, n_copies=150, fam_len=200, div=0.06, seed=5)
    fa = os.path.join(workdir, "grep_emu.fa")
    B.write_fasta(fa, seqs)
    build = emu.dll.jnibwa_createReferenceIndex
    build.argtypes = [ctypes.c_char_p] * 3
    assert build(fa.encode(), fa.encode(), b"auto") == 0 and emu.create_index_file(fa, fa + ".img") == 0
    g = seqs[0][1]
    reads = [bytes(g[st + 20:st + 170]) for st in starts[:3]] + [B.revcomp(bytes(g[starts[5] - 60:starts[5] + 90]))]
    _cmp(emu, oracle, fa + ".img", reads)
    monkeypatch.setenv("BWAMEM_HIP_DEBUGK", "1536")            # 512: the overlap loop by lane; 1024: regions sorted in place
    _cmp(emu, oracle, fa + ".img", reads[:2])


def test_emu_response_block_grows(emu, oracle, small_genome, monkeypatch):
    """the response of jnibwa_createAlignments is sized from the first tiles' bytes per read; when later tiles need more, the
    block is grown (realloc, only while no tile is copying into it) -- here the first tiles are unmappable reads with 8-byte
    records and the later ones real reads, with no slack and several tiles in flight"""
    import random
    seqs, img = small_genome
    rnd = random.Random(8)
    junk = [bytes(rnd.choice(b"ACGT") for _ in range(60)) for _ in range(21)]
    reads = junk + B.simulate_reads(seqs, 40, length=120, seed=12, sub=0.02, indel=0.004)
    monkeypatch.setenv("BWAMEM_HIP_TILE", "7")
    monkeypatch.setenv("BWAMEM_HIP_STREAMS", "3")
    monkeypatch.setenv("BWAMEM_HIP_OUT_SLACK", "0")
    _cmp(emu, oracle, img, reads)


def test_emu_scan_forms(emu, monkeypatch):
    """launch_scan (k_seed.hip): the one-workgroup form, the two-launch form whose blocks add up the block sums before them,
    and the three-launch form with a one-wave scan of the sums must all be numpy's exclusive cumsum."""
    import ctypes
    import numpy as np
    scan = emu.dll._Z11launch_scanPvPKiPliS2_
    scan.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    scan.restype = None
    rng = np.random.default_rng(5)
    for single_max, fused_max, sizes in (("8192", "2048", (0, 1, 63, 4097)), ("1", "2048", (1, 4095, 4096, 3 * 4096 + 17)), ("1", "1", (4097, 70 * 4096 + 5))):
        monkeypatch.setenv("BWAMEM_HIP_SCAN_SINGLE_MAX", single_max)
        monkeypatch.setenv("BWAMEM_HIP_SCAN_FUSED_MAX", fused_max)
        for n in sizes:
            x = rng.integers(0, 1 << 20, size=max(n, 1), dtype=np.int32)[:n]
            x[: n // 2] = rng.integers(0, 2 ** 31 - 1, size=n // 2, dtype=np.int32)       # totals beyond 32 bits
            out = np.full(n + 1, -1, dtype=np.int64)
            tmp = np.zeros(n // 4096 + 4, dtype=np.int64)
            scan(None, x.ctypes.data, out.ctypes.data, n, tmp.ctypes.data)
            want = np.concatenate([[0], np.cumsum(x.astype(np.int64))])
            assert (out == want).all(), (single_max, fused_max, n)


def test_emu_mate_rescue_list_resized(oracle, small_genome):
    """the mate-rescue job list of a tile is sized from earlier tiles; when it does not fit, the plan kernel flags it and the
    alignment and pairing kernels must not run on the half-written list (its unwritten slots hold whatever the memory held).
    Own process: the floor of the list size is read once.  The call is repeated so that the learned size is used as well."""
    import subprocess
    import sys
    B.build_emu()
    seqs, img = small_genome
    code = r'''
import sys
sys.path.insert(0, %r)
import bwalib as B
emu, orc = B.product_lib(emu=True), B.oracle_lib()
seqs = []
for blk in open(%r).read().split(">")[1:]:
    name, _, body = blk.partition("\n")
    seqs.append((name.strip(), body.replace("\n", "").encode()))
pairs = B.simulate_pairs(seqs, 30, length=100, seed=77, ins_mean=300, ins_sd=30, sub=0.05)
for k in range(1, len(pairs), 4):                       # mates that only a rescue can place
    pairs[k] = pairs[k][:30] + B.revcomp(pairs[k][30:70]) + pairs[k][70:]
h, ho = emu.open_index(%r), orc.open_index(%r)
opts = B.set_opt(emu.default_options(), flag=B.MEM_F_PE)
req = B.pack_request(pairs)
want = orc.align_raw(ho, opts, req)
for _ in range(2):
    assert emu.align_raw(h, opts, req) == want
print("rescue-resize-ok")
''' % (os.path.join(B.ROOT, "tests"), img[:-4] if img.endswith(".img") else img, img, img)
    env = dict(os.environ, BWAMEM_HIP_PE_RESCUE_CAP0="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "rescue-resize-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_emu_seed_rescoring_list_resized(oracle, small_genome):
    """the seed re-scoring job list (k_chain.hip: k_rescore_plan) reserves several slots at once, so an overflowing
    reservation leaves slots below the capacity unwritten: the plan kernel must void the list (ERR_RESCUE_CAP), the alignment
    and apply kernels must not touch it, and the tile must be run again with room.  BWAMEM_HIP_RESCORE_CAP0 forces a
    five-slot list on the first attempt.  Own process: the knob is read once."""
    import subprocess
    import sys
    B.build_emu()
    seqs, img = small_genome
    code = r'''
import sys
sys.path.insert(0, %r)
import bwalib as B
emu, orc = B.product_lib(emu=True), B.oracle_lib()
seqs = []
for blk in open(%r).read().split(">")[1:]:
    name, _, body = blk.partition("\n")
    seqs.append((name.strip(), body.replace("\n", "").encode()))
reads = B.simulate_reads(seqs, 2, length=800, seed=5, sub=0.06, indel=0.02)
h, ho = emu.open_index(%r), orc.open_index(%r)
opts = emu.default_options()
req = B.pack_request(reads)
assert emu.align_raw(h, opts, req) == orc.align_raw(ho, opts, req)
print("rescore-resize-ok")
''' % (os.path.join(B.ROOT, "tests"), img[:-4] if img.endswith(".img") else img, img, img)
    env = dict(os.environ, BWAMEM_HIP_RESCORE_CAP0="5")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "rescore-resize-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
