"""Test helpers: ctypes bindings for the jnibwa C ABI (product, emulation build and oracle),
wire-format encode/decode (reference: BwaMemAligner.java:198-307), synthetic genomes and reads
(SURVEY.md section 8(d))."""
import ctypes
import os
import struct
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gatk-bwamem-jni_amd")
HIP_LIB = os.path.join(PKG, "libbwamem_hip.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libbwamem_emu.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

OPT_SIZE = 168
# byte offsets of mem_opt_t fields (BwaMemAligner.java:46-138)
OPT_OFF = dict(a=0, b=4, o_del=8, e_del=12, o_ins=16, e_ins=20, pen_unpaired=24, pen_clip5=28, pen_clip3=32,
               w=36, zdrop=40, max_mem_intv=48, T=56, flag=60, min_seed_len=64, min_chain_weight=68,
               max_chain_extend=72, split_factor=76, split_width=80, max_occ=84, max_chain_gap=88, n_threads=92,
               chunk_size=96, mask_level=100, drop_ratio=104, XA_drop_ratio=108, mask_level_redun=112,
               mapQ_coef_len=116, mapQ_coef_fac=120, max_ins=124, max_matesw=128, max_XA_hits=132,
               max_XA_hits_alt=136, mat=140)
OPT_FLOAT = {"split_factor", "mask_level", "drop_ratio", "XA_drop_ratio", "mask_level_redun", "mapQ_coef_len"}
MEM_F_PE, MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI, MEM_F_NO_RESCUE, MEM_F_PRIMARY5 = 0x2, 0x4, 0x8, 0x10, 0x20, 0x800


def make(target_dir, *args):
    subprocess.run(["make", "-s", "-C", target_dir, *args], check=True)


def build_oracle():
    make(os.path.join(ROOT, "oracle"))
    return ORACLE_LIB


def build_emu():
    make(os.path.join(ROOT, "tests", "emu"))
    return EMU_LIB


class Lib:
    """One library exposing the jnibwa ABI under a symbol prefix ('jnibwa_' or 'oracle_').  A stock libbwa.Linux.so
    (reference build) has the jnibwa_* entry points of jnibwa.h:11-16 but neither of this repo's two helpers: its default
    options come from upstream's mem_opt_init and its buffers are freed with libc free."""

    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.dll = ctypes.CDLL(path)
        libc = ctypes.CDLL(None)

        def f(name):
            if hasattr(self.dll, prefix + name):
                return getattr(self.dll, prefix + name)
            if name == "createDefaultOptions":
                return self.dll.mem_opt_init
            if name == "free":
                return libc.free
            raise AttributeError(prefix + name)
        self._createIndexFile = f("createIndexFile"); self._createIndexFile.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        self._openIndex = f("openIndex"); self._openIndex.restype = ctypes.c_void_p; self._openIndex.argtypes = [ctypes.c_int]
        self._destroyIndex = f("destroyIndex"); self._destroyIndex.argtypes = [ctypes.c_void_p]
        self._names = f("getRefContigNames"); self._names.restype = ctypes.c_void_p
        self._names.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        self._align = f("createAlignments"); self._align.restype = ctypes.c_void_p
        self._align.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        self._opts = f("createDefaultOptions"); self._opts.restype = ctypes.c_void_p
        self._free = f("free"); self._free.argtypes = [ctypes.c_void_p]

    def create_index_file(self, prefix, img):
        return self._createIndexFile(prefix.encode(), img.encode())

    def open_index(self, img):
        fd = os.open(img, os.O_RDONLY)
        h = self._openIndex(fd)
        return h

    def destroy_index(self, h):
        return self._destroyIndex(h)

    def contig_names(self, h):
        sz = ctypes.c_size_t()
        p = self._names(h, ctypes.byref(sz))
        buf = ctypes.string_at(p, sz.value)
        self._free(p)
        n, = struct.unpack_from("<i", buf, 0)
        off, out = 4, []
        for _ in range(n):
            l, = struct.unpack_from("<i", buf, off); off += 4
            out.append(buf[off:off + l].decode()); off += l
        return out

    def default_options(self):
        p = self._opts()
        b = ctypes.string_at(p, OPT_SIZE)
        self._free(p)
        return bytearray(b)

    def align_raw(self, h, opts, request, pes=None):
        """-> raw response bytes, or None when the library returned NULL."""
        ob = ctypes.create_string_buffer(bytes(opts), OPT_SIZE)
        rb = ctypes.create_string_buffer(request, len(request))
        pb = ctypes.create_string_buffer(pes, len(pes)) if pes is not None else None
        sz = ctypes.c_size_t()
        p = self._align(h, ob, pb, rb, ctypes.byref(sz))
        if not p:
            return None
        out = ctypes.string_at(p, sz.value)
        self._free(p)
        return out


def product_lib(emu=False):
    return Lib(EMU_LIB if emu else HIP_LIB, "jnibwa_")


def oracle_lib():
    return Lib(ORACLE_LIB, "oracle_")


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def stock_libbwa():
    """The hooks that turn "parity unpinned" into "pinned" the moment the material exists:
      LIBBWA_PATH     a stock libbwa.Linux.so as the reference's own Makefile builds it (src/main/c/Makefile:22-23), or
      BWA_ORACLE_SRC  a checkout of lh3/bwa at cb950614, compiled here together with the reference's jnibwa.c where it lies
                      under /root/reference (recipe: oracle/Makefile target `ref`, output oracle/_ref/libbwa_ref.so)
    -> a Lib behind the same jnibwa_* ABI, used as a second checker by every parity test and as bench.py's CPU baseline
    (cpu_baseline.kind = "reference"); None when neither is set (this offline image has no bwa source)."""
    path = os.environ.get("LIBBWA_PATH")
    if not path and os.environ.get("BWA_ORACLE_SRC"):
        make(os.path.join(ROOT, "oracle"), "ref", "BWA_ORACLE_SRC=" + os.environ["BWA_ORACLE_SRC"])
        path = os.path.join(REF_DIR, "libbwa_ref.so")
    if not path:
        return None
    if not os.path.exists(path):
        raise FileNotFoundError("LIBBWA_PATH / BWA_ORACLE_SRC is set but %s does not exist" % path)
    return Lib(path, "jnibwa_")


def check_against_stock(img, opts, request, got, pes=None):
    """second checker: when a stock libbwa is available, `got` must also equal its response (no-op otherwise)"""
    ref = stock_libbwa()
    if ref is None:
        return False
    h = ref.open_index(img)
    try:
        want = ref.align_raw(h, opts, request, pes)
    finally:
        ref.destroy_index(h)
    assert got == want, "response differs from the stock libbwa at %s" % ref.path
    return True


def set_opt(opts, **kw):
    for k, v in kw.items():
        if k == "mat":
            opts[140:165] = bytes((x & 0xff) for x in v)
        elif k == "max_mem_intv":
            struct.pack_into("<q", opts, OPT_OFF[k], v)
        elif k in OPT_FLOAT:
            struct.pack_into("<f", opts, OPT_OFF[k], v)
        else:
            struct.pack_into("<i", opts, OPT_OFF[k], v)
    return opts


def get_opt(opts, k):
    if k == "max_mem_intv":
        return struct.unpack_from("<q", opts, OPT_OFF[k])[0]
    return struct.unpack_from("<f" if k in OPT_FLOAT else "<i", opts, OPT_OFF[k])[0]


def pack_request(seqs):
    """BwaMemAligner.java:198-209: int32 count, then NUL-terminated base strings."""
    parts = [struct.pack("<i", len(seqs))]
    for s in seqs:
        parts.append(s if isinstance(s, bytes) else s.encode())
        parts.append(b"\0")
    return b"".join(parts)


def pack_pestat(low, high, avg, std, failed=False):
    """...BwaMemIndex.c:21-40: only orientation slot 1 (FR) can be supplied; the others fail."""
    b = b""
    for i in range(4):
        if i == 1:
            b += struct.pack("<iiiidd", low if not failed else 0, high if not failed else 0, 1 if failed else 0, 0,
                             avg if not failed else 0.0, std if not failed else 0.0)
        else:
            b += struct.pack("<iiiidd", 0, 0, 1, 0, 0.0, 0.0)
    return b


CIGAR_OPS = "MID?S???????????"


def decode_response(buf, n_reads):
    """BwaMemAligner.java:215-307."""
    off, out = 0, []
    for _ in range(n_reads):
        na, = struct.unpack_from("<i", buf, off); off += 4
        alns = []
        for _ in range(na):
            fm, = struct.unpack_from("<i", buf, off); off += 4
            flag, mapq = (fm >> 16) & 0xffff, fm & 0xff
            d = dict(flag=flag, mapq=mapq)
            if not flag & 4:
                rid, pos, nm, AS, XS, nc = struct.unpack_from("<6i", buf, off); off += 24
                cig = struct.unpack_from("<%di" % nc, buf, off); off += 4 * nc
                nmd, = struct.unpack_from("<i", buf, off); off += 4
                md = buf[off:off + nmd]; off += (nmd + 3) & ~3
                nxa, = struct.unpack_from("<i", buf, off); off += 4
                xa = buf[off:off + nxa]; off += (nxa + 3) & ~3
                cigar = "".join("%d%s" % (c >> 4, CIGAR_OPS[c & 15]) for c in cig)
                ref_len = sum(c >> 4 for c in cig if CIGAR_OPS[c & 15] in "MD")
                seq_start = (cig[0] >> 4) if nc and CIGAR_OPS[cig[0] & 15] == "S" else 0
                seq_len = sum(c >> 4 for c in cig if CIGAR_OPS[c & 15] in "MI")
                d.update(rid=rid, pos=pos, ref_end=pos + ref_len, seq_start=seq_start, seq_end=seq_start + seq_len,
                         nm=nm, AS=AS, XS=XS, cigar=cigar, md=md.decode(), xa=xa.decode())
            if flag & 9 == 1:
                mr, mp, tl = struct.unpack_from("<3i", buf, off); off += 12
                d.update(mrid=mr, mpos=mp, tlen=tl)
            alns.append(d)
        out.append(alns)
    assert off == len(buf), (off, len(buf))
    return out


def split_response(buf, n_reads):
    """per-read byte slices of a response (the reference's bufLen walk, jnibwa.c:99-124)."""
    off, out = 0, []
    for _ in range(n_reads):
        start = off
        na, = struct.unpack_from("<i", buf, off); off += 4
        for _ in range(na):
            fm, = struct.unpack_from("<i", buf, off); off += 4
            flag = (fm >> 16) & 0xffff
            if not flag & 4:
                nc, = struct.unpack_from("<i", buf, off + 20); off += 24 + 4 * nc
                nmd, = struct.unpack_from("<i", buf, off); off += 4 + ((nmd + 3) & ~3)
                nxa, = struct.unpack_from("<i", buf, off); off += 4 + ((nxa + 3) & ~3)
            if flag & 9 == 1:
                off += 12
        out.append(buf[start:off])
    assert off == len(buf)
    return out


# ---------------------------------------------------------------- synthetic data
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}


def revcomp(s):
    return bytes(COMP[c] for c in reversed(s))


def synth_genome(total_bp, n_contigs=3, seed=0x5EED, repeat_frac=0.05, n_frac=0.0):
    """i.i.d. ACGT contigs; repeat_frac of the bases overwritten by diverged copies of 300bp-6kb segments."""
    rng = np.random.default_rng(seed)
    lens = np.maximum(200, (rng.dirichlet(np.ones(n_contigs) * 4) * total_bp).astype(np.int64))
    g = rng.integers(0, 4, size=int(lens.sum()), dtype=np.uint8)
    n_rep = int(repeat_frac * len(g))
    done = 0
    while done < n_rep and len(g) > 1000:
        l = int(rng.integers(300, min(6000, len(g) // 4)))
        src = int(rng.integers(0, len(g) - l)); dst = int(rng.integers(0, len(g) - l))
        seg = g[src:src + l].copy()
        mut = rng.random(l) < 0.03
        seg[mut] = (seg[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
        if rng.random() < 0.5:
            seg = (3 - seg)[::-1]
        g[dst:dst + l] = seg
        done += l
    seqs, off = [], 0
    for i, l in enumerate(lens):
        s = BASES[g[off:off + l]].copy()
        if n_frac > 0:
            k = int(n_frac * l)
            if k:
                st = int(rng.integers(0, l - k)); s[st:st + k] = ord("N")
        seqs.append(("chr%d" % (i + 1), s.tobytes()))
        off += l
    return seqs


def write_fasta(path, seqs, width=60):
    with open(path, "wb") as f:
        for name, s in seqs:
            f.write(b">" + name.encode() + b"\n")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + b"\n")


def simulate_reads(seqs, n, length=150, seed=42, sub=0.01, indel=0.0002, n_rate=0.001, random_frac=0.005):
    """single-end reads: uniform position/strand, substitutions, geometric-length indels, N bases, some random reads."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(s) for _, s in seqs], dtype=np.float64)
    reads = []
    for _ in range(n):
        if rng.random() < random_frac:
            reads.append(BASES[rng.integers(0, 4, size=length)].tobytes())
            continue
        while True:
            ci = int(rng.choice(len(seqs), p=lens / lens.sum()))
            s = seqs[ci][1]
            if len(s) > length + 50:
                break
        st = int(rng.integers(0, len(s) - length - 40))
        src = bytearray(s[st:st + length + 40])
        out = bytearray()
        i = 0
        while len(out) < length and i < len(src):
            r = rng.random()
            if r < indel / 2:
                i += int(rng.geometric(0.5))                       # deletion
                continue
            if r < indel:
                out += BASES[rng.integers(0, 4, size=int(rng.geometric(0.5)))].tobytes()   # insertion
                continue
            c = src[i]; i += 1
            if c != 78 and rng.random() < sub:
                c = int(BASES[(list(b"ACGT").index(c) + int(rng.integers(1, 4))) % 4])
            if rng.random() < n_rate:
                c = 78
            out.append(c)
        rd = bytes(out[:length])
        if rng.random() < 0.5:
            rd = revcomp(rd)
        reads.append(rd)
    return reads


def simulate_pairs(seqs, n_pairs, length=150, seed=43, ins_mean=400, ins_sd=50, **kw):
    rng = np.random.default_rng(seed)
    lens = np.array([len(s) for _, s in seqs], dtype=np.float64)
    out = []
    for _ in range(n_pairs):
        ci = int(rng.choice(len(seqs), p=lens / lens.sum()))
        s = seqs[ci][1]
        isz = int(np.clip(rng.normal(ins_mean, ins_sd), length, 1000))
        if len(s) <= isz + 10:
            isz = len(s) - 10
        st = int(rng.integers(0, len(s) - isz))
        frag = bytearray(s[st:st + isz])
        for j in range(len(frag)):
            if frag[j] != 78 and rng.random() < kw.get("sub", 0.01):
                frag[j] = int(BASES[(list(b"ACGT").index(frag[j]) + int(rng.integers(1, 4))) % 4])
        r1, r2 = bytes(frag[:length]), revcomp(bytes(frag[-length:]))
        if rng.random() < 0.5:
            r1, r2 = r2, r1
        out += [r1, r2]
    return out


def _diverge(rng, seg, sub, indel=0.0):
    """a diverged copy of a base-code array: substitutions plus a few short indels"""
    seg = seg.copy()
    mut = rng.random(len(seg)) < sub
    seg[mut] = (seg[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
    if indel <= 0:
        return seg
    out, i = [], 0
    cuts = np.flatnonzero(rng.random(len(seg)) < indel)
    for c in cuts:
        out.append(seg[i:c])
        if rng.random() < 0.5:
            i = c + int(rng.integers(1, 6))                                   # deletion
        else:
            out.append(rng.integers(0, 4, size=int(rng.integers(1, 6)), dtype=np.uint8)); i = c   # insertion
    out.append(seg[i:])
    return np.concatenate(out)


def synth_alt_genome(seed=0xA17):
    """A primary assembly of four contigs plus ALT contigs the way GRCh38 carries them:
    chr1_alt1 / chr2_alt1 / chr1_alt2 are 2-5 % diverged copies of 20-35 kb of chr1 / chr2 (one reverse-complemented, two
    overlapping the same source), chrUn_decoy is unrelated sequence listed as ALT, and a 500-base repeat family with seven
    primary copies also sits inside an ALT source region, so that some reads have more hits than max_XA_hits of which one is ALT.
    -> (seqs, alt_names, regions) with regions = {name: (contig index, start, end)} of the stretches reads should be drawn from."""
    rng = np.random.default_rng(seed)
    base = [("chr%d" % (i + 1), None) for i in range(4)]
    arr = [rng.integers(0, 4, size=l, dtype=np.uint8) for l in (110000, 80000, 60000, 50000)]
    fam = arr[0][60000:60500].copy()                                           # the repeat family, founder inside alt1's source
    spots = [(0, 5000), (1, 3000), (1, 30000), (2, 2000), (2, 20000), (3, 1000), (3, 9000)]
    for ci, st in spots:
        if st + 500 < len(arr[ci]):
            arr[ci][st:st + 500] = _diverge(rng, fam, 0.015)
    alts = [("chr1_alt1", _diverge(rng, arr[0][40000:75000], 0.03, 0.0005)),
            ("chr2_alt1", (3 - _diverge(rng, arr[1][10000:35000], 0.02, 0.0005))[::-1]),
            ("chr1_alt2", _diverge(rng, arr[0][55000:75000], 0.05, 0.001)),
            ("chrUn_decoy", rng.integers(0, 4, size=8000, dtype=np.uint8))]
    seqs = [(n, BASES[a].tobytes()) for (n, _), a in zip(base, arr)] + [(n, BASES[a].tobytes()) for n, a in alts]
    regions = {"chr1_src": (0, 40000, 75000), "chr2_src": (1, 10000, 35000), "family": (0, 60000, 60500),
               "chr1_alt1": (4, 0, len(alts[0][1])), "chr2_alt1": (5, 0, len(alts[1][1])), "chr1_alt2": (6, 0, len(alts[2][1])),
               "decoy": (7, 0, 8000)}
    return seqs, [n for n, _ in alts], regions


def write_alt_file(path, alt_names):
    """<prefix>.alt as bwa-kit ships it: SAM text, '@' header lines, the ALT contig's name in the first column"""
    with open(path, "w") as f:
        f.write("@HD\tVN:1.5\tSO:unsorted\n@SQ\tSN:chr1\tLN:1\n")
        for i, n in enumerate(alt_names):
            f.write("%s\t%d\tchr1\t%d\t60\t100M\t*\t0\t0\t*\t*\tNM:i:0\n" % (n, 0 if i % 2 == 0 else 16, 1000 * i + 1))


def reads_from_regions(seqs, regions, names, n, length=150, seed=1, **kw):
    """simulate_reads restricted to the named regions"""
    sub = [(k, seqs[regions[k][0]][1][regions[k][1]:regions[k][2]]) for k in names]
    return simulate_reads(sub, n, length=length, seed=seed, random_frac=0.0, **kw)


def pairs_from_regions(seqs, regions, names, n_pairs, length=100, seed=1, **kw):
    sub = [(k, seqs[regions[k][0]][1][regions[k][1]:regions[k][2]]) for k in names]
    return simulate_pairs(sub, n_pairs, length=length, seed=seed, **kw)


def synth_repeat_genome(total_bp=1500000, n_copies=1200, fam_len=300, div=0.10, seed=0xA1B):
    """A genome with one SINE-like family: n_copies copies of a fam_len-base consensus at `div` divergence scattered over
    i.i.d. sequence (both strands).  Reads drawn from the copies seed in hundreds of places (max_occ sampling, re-seeding,
    frac_rep) and carry hundreds of chains into mem_chain_flt -- what a read in an Alu does on GRCh38.
    -> (seqs, starts of the copies in contig 0)"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=total_bp, dtype=np.uint8)
    cons = rng.integers(0, 4, size=fam_len, dtype=np.uint8)
    starts = np.sort(rng.choice(np.arange(1000, total_bp - fam_len - 1000, fam_len + 50), size=n_copies, replace=False))
    for st in starts:
        c = _diverge(rng, cons, div)
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        g[st:st + fam_len] = c
    return [("chrR", BASES[g].tobytes())], [int(x) for x in starts]


def synth_tandem_genome(total_bp=400000, seed=0x7A4D, n_arrays=3, array_bp=30000, mono_len=171, hor=4, div=0.015, n_sine=250):
    """Repeat structure that makes paired-end finalisation heavy (what centromeric satellites and Alus do on GRCh38): tandem
    arrays of a `mono_len`-base monomer in `hor`-monomer higher-order repeats (copies `div` diverged), plus a dispersed
    300-base family.  A pair inside an array has hundreds of hits per end, dozens of rescue anchors, and rescued
    regions that tie with existing hits in score and end coordinate -- the cases mem_matesw's repeated
    mem_sort_dedup_patch decides by sort order.  -> (seqs, [(start, end) of the arrays in contig 0])"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=total_bp, dtype=np.uint8)
    mono = rng.integers(0, 4, size=mono_len, dtype=np.uint8)
    unit = np.concatenate([_diverge(rng, mono, 0.2) for _ in range(hor)])
    arrays = []
    for k in range(n_arrays):
        st = 20000 + k * (total_bp - 40000) // n_arrays
        arr = unit[np.arange(array_bp) % len(unit)]
        g[st:st + array_bp] = _diverge(rng, arr, div)
        arrays.append((st, st + array_bp))
    cons = rng.integers(0, 4, size=300, dtype=np.uint8)
    for _ in range(n_sine):
        st = int(rng.integers(1000, total_bp - 1400))
        if any(a - 400 < st < b for a, b in arrays):
            continue
        c = _diverge(rng, cons, 0.10)
        g[st:st + 300] = (3 - c)[::-1] if rng.random() < 0.5 else c
    return [("chrT", BASES[g].tobytes())], arrays
