#!/usr/bin/env python3
"""bench.py -- reads aligned per second on MI355X for BASELINE.json's headline workload.

Workload (config.workload): configs[1] of BASELINE.json -- 10 M x 150 bp single-end synthetic
reads vs a GRCh38-scale reference, full index resident in HBM.  GRCh38 itself is not available
offline, so the reference is the seeded synthetic genome of SURVEY.md section 8(d): 24 contigs,
3.1e9 bp, i.i.d. ACGT with 5 % of the bases overwritten by diverged copies of 300 bp - 6 kb
segments; it is indexed on the device by gatk-bwamem-jni_amd/index_build_gpu.py.

A "step" is one pass of the whole hot path (encode -> SMEM seeding -> SA lookup -> chaining ->
banded extension -> region post-processing -> records -> packed response) over the batch, with
the request already resident in HBM and the response left in HBM: that rate is `value`.  After
the timed steps the same batch goes through the drop-in entry point itself,
jnibwa_createAlignments (pageable host request in, malloc'ed host response out): `host_to_host`.
One process per GPU; `--gpus N` starts the N ranks itself when no launcher did; reads shard
across ranks with no collective (weak scaling: every rank aligns --reads reads); the only
torch.distributed traffic is the barriers and the gather of the ranks' elapsed times.
--genome humanlike swaps the i.i.d. reference for one with human-like repeat content; --image /
BWAHIP_REF_IMG uses an existing index image (reads sampled from its packed reference);
LIBBWA_PATH / BWA_ORACLE_SRC make a stock libbwa the CPU baseline and parity checker.
"""
import argparse
import ctypes
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gatk-bwamem-jni_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("n_reads", "n_ext", "n_lf", "n_sa", "n_dp_cells")] + \
               [(n, ctypes.c_double) for n in ("ms_encode", "ms_seed", "ms_sa", "ms_chain", "ms_extend", "ms_post", "ms_final", "ms_pack", "ms_other")] + \
               [(n, ctypes.c_uint64) for n in ("n_launch_seed", "n_launch_sa", "n_launch_extend", "n_tiles", "n_retries")]


def load_lib():
    path = os.path.join(PKG, "libbwamem_hip.so")
    if not os.path.exists(path):
        raise SystemExit("libbwamem_hip.so is missing (run __graft_entry__.build()); there is no fallback path")
    lib = ctypes.CDLL(path)
    lib.jnibwa_openIndex.restype = ctypes.c_void_p
    lib.jnibwa_openIndex.argtypes = [ctypes.c_int]
    lib.jnibwa_destroyIndex.argtypes = [ctypes.c_void_p]
    lib.jnibwa_createDefaultOptions.restype = ctypes.c_void_p
    lib.jnibwa_createAlignments.restype = ctypes.c_void_p
    lib.jnibwa_createAlignments.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.jnibwa_free.argtypes = [ctypes.c_void_p]
    lib.bwamem_hip_set_device.argtypes = [ctypes.c_int]
    lib.bwamem_hip_index_replicas.argtypes = [ctypes.c_void_p]
    lib.bwamem_hip_build_image.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    lib.bwamem_hip_batch_wrap_device.restype = ctypes.c_void_p
    lib.bwamem_hip_batch_wrap_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p]
    lib.bwamem_hip_batch_align.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    lib.bwamem_hip_batch_result_bytes.restype = ctypes.c_size_t
    lib.bwamem_hip_batch_result_bytes.argtypes = [ctypes.c_void_p]
    lib.bwamem_hip_batch_download.restype = ctypes.c_int
    lib.bwamem_hip_batch_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.bwamem_hip_batch_free.argtypes = [ctypes.c_void_p]
    lib.bwamem_hip_stats_get.argtypes = [ctypes.POINTER(Stats)]
    return lib


def synth_genome(torch, dev, total_bp, n_contigs, seed):
    """i.i.d. ACGT + 5 % diverged repeat copies; returns (codes uint8, [(name, len)])"""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    w = torch.rand(n_contigs, generator=g, device=dev) + 0.5
    lens = (w / w.sum() * total_bp).long()
    lens[-1] += total_bp - int(lens.sum())
    codes = torch.empty(total_bp, dtype=torch.uint8, device=dev)
    for lo in range(0, total_bp, 1 << 30):                       # some torch ops are limited to < 2^31 elements per call
        hi = min(total_bp, lo + (1 << 30))
        codes[lo:hi] = torch.randint(0, 4, (hi - lo,), dtype=torch.uint8, generator=g, device=dev)
    n_rep = int(0.05 * total_bp)
    hg = torch.Generator(); hg.manual_seed(seed + 1)
    done = 0
    while done < n_rep and total_bp > 20000:
        l = int(torch.randint(300, 6000, (1,), generator=hg))
        src = int(torch.randint(0, total_bp - l, (1,), generator=hg)); dst = int(torch.randint(0, total_bp - l, (1,), generator=hg))
        seg = codes[src:src + l].clone()
        mut = torch.rand(l, generator=g, device=dev) < 0.03
        seg = torch.where(mut, (seg + torch.randint(1, 4, (l,), dtype=torch.uint8, generator=g, device=dev)) % 4, seg)
        if int(torch.randint(0, 2, (1,), generator=hg)):
            seg = torch.flip(3 - seg, [0])
        codes[dst:dst + l] = seg
        done += l
    contigs = [("chr%d" % (i + 1), int(lens[i])) for i in range(n_contigs)]
    return codes, contigs


def genome_from_index(torch, dev, lib, idx):
    """--image / BWAHIP_REF_IMG: base codes and contig table of an existing index image, taken from its packed reference
    on the device (ambiguous stretches hold whatever random bases the indexer put there, as in any .pac)"""
    lib.bwamem_hip_index_contig_lengths.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.bwamem_hip_index_unpack_pac.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    n = lib.bwamem_hip_index_contig_lengths(idx, None, 0)
    lens = (ctypes.c_int64 * max(n, 1))()
    lib.bwamem_hip_index_contig_lengths(idx, lens, n)
    total = sum(lens[i] for i in range(n))
    codes = torch.empty(total, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    if lib.bwamem_hip_index_unpack_pac(idx, 0, total, codes.data_ptr()) != 0:
        raise SystemExit("cannot unpack the image's reference")
    return codes, [("ctg%d" % i, int(lens[i])) for i in range(n)]


def _mutate(torch, g, dev, seg, rate):
    """substitutions at a per-row rate (rate: scalar or (rows, 1) tensor)"""
    mut = torch.rand(seg.shape, generator=g, device=dev) < rate
    return torch.where(mut, (seg + torch.randint(1, 4, seg.shape, dtype=torch.uint8, generator=g, device=dev)) % 4, seg)


def synth_genome_humanlike(torch, dev, total_bp, n_contigs, seed):
    """A harder reference than synth_genome: about 45 % of the bases belong to repeat families laid out the way a human
    genome has them, so that seeds hit many places, max_occ sampling and frac_rep come into play, and reads carry several
    chains.  Fractions scale with total_bp (figures for 3.1 Gbp):
      * SINE-like: one 300-base consensus with an A-rich tail, ~1.1 M copies at 8-16 % divergence           (~10.5 %)
      * LINE-like: one 6-kb consensus, ~0.5 M copies, 5'-truncated (5 % full length), 3-20 % divergence      (~17 %)
      * older dispersed families: 40 consensi of 200-900 bases, 20-30 % divergence                           (~8 %)
      * satellite arrays: one 171-base monomer in 12-monomer higher-order repeats, 1-3 Mbp per contig, 1.5 % (~1.5 %)
      * microsatellites / low complexity: 1-6 base units, 20-80 bases                                        (~1 %)
      * segmental duplications: 10-100 kb blocks copied at 1-3 % divergence                                  (~5 %)
    over i.i.d. bases with 41 % GC.  Seeded and device-side like synth_genome."""
    g = torch.Generator(device=dev); g.manual_seed(seed ^ 0x48554D)
    hg = torch.Generator(); hg.manual_seed(seed + 7)
    w = torch.rand(n_contigs, generator=g, device=dev) + 0.5
    lens = (w / w.sum() * total_bp).long()
    lens[-1] += total_bp - int(lens.sum())
    codes = torch.empty(total_bp, dtype=torch.uint8, device=dev)
    lut = torch.tensor([0] * 295 + [1] * 205 + [2] * 205 + [3] * 295, dtype=torch.uint8, device=dev)     # A/T 29.5 %, C/G 20.5 %
    for lo in range(0, total_bp, 1 << 28):
        hi = min(total_bp, lo + (1 << 28))
        codes[lo:hi] = lut[torch.randint(0, 1000, (hi - lo,), generator=g, device=dev)]
    scale = total_bp / 3.1e9

    def rand_consensus(n):
        return lut[torch.randint(0, 1000, (n,), generator=g, device=dev)]

    def scatter_family(cons, n_copies, div_lo, div_hi, min_len, full_frac, chunk):
        """copies of cons (3' end kept, 5' end truncated to a random length >= min_len unless full) at random places and strands"""
        Lc = cons.numel()
        col = torch.arange(Lc, device=dev)
        for c0 in range(0, n_copies, chunk):
            m = min(chunk, n_copies - c0)
            ln = torch.where(torch.rand(m, generator=g, device=dev) < full_frac, torch.full((m,), Lc, device=dev),
                             (torch.rand(m, generator=g, device=dev) * (Lc - min_len)).long() + min_len)
            seg = _mutate(torch, g, dev, cons[None, :].expand(m, Lc), (torch.rand(m, 1, generator=g, device=dev) * (div_hi - div_lo) + div_lo))
            rc = torch.rand(m, generator=g, device=dev) < 0.5
            seg = torch.where(rc[:, None], torch.flip(3 - seg, [1]), seg)
            keep = torch.where(rc[:, None], col[None, :] < ln[:, None], col[None, :] >= (Lc - ln)[:, None])      # the kept end
            pos = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (total_bp - Lc - 1)).long()
            dst = pos[:, None] + col[None, :]
            codes[dst[keep]] = seg[keep]

    # older dispersed families first (younger ones overwrite them, as insertions do)
    for _ in range(40):
        Lc = int(torch.randint(200, 900, (1,), generator=hg))
        scatter_family(rand_consensus(Lc), int(0.08 * total_bp / 40 / (0.75 * Lc)), 0.20, 0.30, Lc // 2, 0.5, 1 << 16)
    line = rand_consensus(6000)
    scatter_family(line, int(0.17 * total_bp / 1050), 0.03, 0.20, 300, 0.05, 1 << 13)
    sine = rand_consensus(300)
    sine[-24:] = 0                                                                  # A-rich tail
    scatter_family(sine, int(0.105 * total_bp / 295), 0.08, 0.16, 250, 0.9, 1 << 17)
    # microsatellites / low complexity
    n_ms = int(0.01 * total_bp / 50)
    col = torch.arange(80, device=dev)
    for c0 in range(0, n_ms, 1 << 18):
        m = min(1 << 18, n_ms - c0)
        ul = torch.randint(1, 7, (m,), generator=g, device=dev)
        unit = torch.randint(0, 4, (m, 6), dtype=torch.uint8, generator=g, device=dev)
        ln = torch.randint(20, 81, (m,), generator=g, device=dev)
        seg = torch.gather(unit, 1, (col[None, :] % ul[:, None]))
        seg = _mutate(torch, g, dev, seg, 0.02)
        pos = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (total_bp - 100)).long()
        keep = col[None, :] < ln[:, None]
        codes[(pos[:, None] + col[None, :])[keep]] = seg[keep]
    # satellite arrays: one per contig
    mono = rand_consensus(171)
    hor = torch.cat([_mutate(torch, g, dev, mono, 0.2) for _ in range(12)])
    off = 0
    for i in range(n_contigs):
        cl = int(lens[i])
        al = int(min(cl // 4, (1_000_000 + int(torch.randint(0, 2_000_000, (1,), generator=hg))) * min(1.0, scale * 1.0 + 0.0)))
        if al > 4000:
            st = off + int(torch.randint(0, max(1, cl - al), (1,), generator=hg))
            arr = hor[torch.arange(al, device=dev) % hor.numel()]
            codes[st:st + al] = _mutate(torch, g, dev, arr, 0.015)
        off += cl
    # segmental duplications
    n_sd = int(0.05 * total_bp)
    done = 0
    while done < n_sd and total_bp > 400000:
        l = int(torch.randint(10000, min(100000, total_bp // 8), (1,), generator=hg))
        src = int(torch.randint(0, total_bp - l, (1,), generator=hg)); dst = int(torch.randint(0, total_bp - l, (1,), generator=hg))
        seg = _mutate(torch, g, dev, codes[src:src + l].clone(), 0.01 + 0.02 * float(torch.rand(1, generator=hg)))
        if int(torch.randint(0, 2, (1,), generator=hg)):
            seg = torch.flip(3 - seg, [0])
        codes[dst:dst + l] = seg
        done += l
    contigs = [("chr%d" % (i + 1), int(lens[i])) for i in range(n_contigs)]
    return codes, contigs


def synth_pairs(torch, dev, codes, contigs, n_pairs, length, seed):
    """config 2 of BASELINE.json (cfg 3 there): FR pairs, insert size ~ N(400, 50^2) clipped to [length, 1000]; mates interleaved as Java sends them."""
    g = torch.Generator(device=dev); g.manual_seed(seed ^ 0x9E37)
    total = codes.numel()
    span = 1040 + length
    bounds = torch.tensor([0] + [l for _, l in contigs], device=dev).cumsum(0)
    pos1 = (torch.rand(n_pairs, generator=g, device=dev, dtype=torch.float64) * (total - span)).long()
    ci = torch.searchsorted(bounds, pos1, right=True) - 1
    pos1 = torch.where(pos1 + span <= bounds[ci + 1], pos1, torch.clamp(bounds[ci + 1] - span, min=0))
    isize = torch.clamp((torch.randn(n_pairs, generator=g, device=dev) * 50 + 400).long(), min=length, max=1000)
    pos2 = pos1 + isize - length
    fs = torch.rand(n_pairs, generator=g, device=dev) < 0.5
    r1 = synth_reads(torch, dev, codes, contigs, n_pairs, length, seed, pos=torch.where(fs, pos1, pos2), rc=~fs)
    r2 = synth_reads(torch, dev, codes, contigs, n_pairs, length, seed + 1, pos=torch.where(fs, pos2, pos1), rc=fs)
    return torch.stack([r1, r2], dim=1).reshape(2 * n_pairs, length + 1).contiguous()


def synth_reads(torch, dev, codes, contigs, n, length, seed, pos=None, rc=None):
    """uniform position/strand, 1 % substitutions, 0.02 %/base indels (one event per affected
    read, geometric length), 0.1 % N, 0.5 % random reads.  Returns the request payload (n x (length+1) ASCII, NUL-terminated)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    total = codes.numel()
    slack = 40
    # start positions that keep length+slack bases inside one contig
    bounds = torch.tensor([0] + [l for _, l in contigs], device=dev).cumsum(0)
    if pos is None:
        pos = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (total - length - slack)).long()
        ci = torch.searchsorted(bounds, pos, right=True) - 1
        end_ok = pos + length + slack <= bounds[ci + 1]
        pos = torch.where(end_ok, pos, torch.clamp(bounds[ci + 1] - length - slack, min=0))
    col = torch.arange(length, device=dev)
    # one indel event for a fraction of reads
    has = torch.rand(n, generator=g, device=dev) < (0.0002 * length)
    is_del = torch.rand(n, generator=g, device=dev) < 0.5
    ev = (torch.rand(n, generator=g, device=dev) * (length - 20)).long() + 10
    k = torch.clamp(torch.log(torch.rand(n, generator=g, device=dev)).div(-0.6931).floor().long() + 1, max=slack - 1)
    k = torch.where(has, k, torch.zeros_like(k))
    shift = torch.where(is_del, k, -k)                                   # deletion reads further right, insertion shifts left
    after = col[None, :] >= torch.where(is_del, ev, ev + k)[:, None]
    src = pos[:, None] + col[None, :] + torch.where(after, shift[:, None], torch.zeros_like(shift)[:, None])
    b = codes[src]
    ins_zone = (~is_del)[:, None] & (col[None, :] >= ev[:, None]) & (col[None, :] < (ev + k)[:, None])
    rnd = torch.randint(0, 4, (n, length), dtype=torch.uint8, generator=g, device=dev)
    b = torch.where(ins_zone, rnd, b)
    sub = torch.rand(n, length, generator=g, device=dev) < 0.01
    b = torch.where(sub, (b + 1 + rnd % 3) % 4, b)
    if rc is None:
        rc = torch.rand(n, generator=g, device=dev) < 0.5
    b = torch.where(rc[:, None], torch.flip(3 - b, [1]), b)
    junk = torch.rand(n, generator=g, device=dev) < 0.005
    b = torch.where(junk[:, None], rnd, b)
    asc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[b.long()]
    nmask = torch.rand(n, length, generator=g, device=dev) < 0.001
    asc = torch.where(nmask, torch.full_like(asc, ord("N")), asc)
    payload = torch.zeros((n, length + 1), dtype=torch.uint8, device=dev)
    payload[:, :length] = asc
    return payload


def synth_long_reads(torch, dev, codes, contigs, n, length, seed):
    """config 5 of BASELINE.json: ONT-style reads, 8 % substitutions, 3 % inserted and 3 % deleted bases
    (single-base events), uniform position/strand.  Generated in chunks: the per-base index arithmetic is 8 bytes a base."""
    g = torch.Generator(device=dev); g.manual_seed(seed ^ 0x0117)
    total = codes.numel()
    slack = int(length * 0.08) + 64
    bounds = torch.tensor([0] + [l for _, l in contigs], device=dev).cumsum(0)
    asc_tab = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    payload = torch.zeros((n, length + 1), dtype=torch.uint8, device=dev)
    step = max(1, min(n, 200_000_000 // max(length, 1)))
    for c0 in range(0, n, step):
        m = min(step, n - c0)
        pos = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (total - length - slack)).long()
        ci = torch.searchsorted(bounds, pos, right=True) - 1
        pos = torch.where(pos + length + slack <= bounds[ci + 1], pos, torch.clamp(bounds[ci + 1] - length - slack, min=0))
        ins = torch.rand(m, length, generator=g, device=dev) < 0.03                 # this output base is an inserted one
        dele = torch.rand(m, length, generator=g, device=dev) < 0.03                # one reference base is skipped after this one
        adv = torch.where(ins, torch.zeros((), dtype=torch.int64, device=dev), 1 + dele.long())
        src = pos[:, None] + torch.cumsum(adv, 1) - adv                             # exclusive prefix: reference index of column j
        src = torch.minimum(src, (pos + length + slack - 1)[:, None])
        b = codes[src]
        rnd = torch.randint(0, 4, (m, length), dtype=torch.uint8, generator=g, device=dev)
        b = torch.where(ins, rnd, b)
        sub = (torch.rand(m, length, generator=g, device=dev) < 0.08) & ~ins
        b = torch.where(sub, (b + 1 + rnd % 3) % 4, b)
        rc = torch.rand(m, generator=g, device=dev) < 0.5
        b = torch.where(rc[:, None], torch.flip(3 - b, [1]), b)
        payload[c0:c0 + m, :length] = asc_tab[b.long()]
        del ins, dele, adv, src, b, rnd, sub
    return payload


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without a launcher: this process starts them itself")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome-bp", type=int, default=3_100_000_000)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--genome", choices=["iid", "humanlike"], default="iid", help="iid: i.i.d. bases + 5 %% diverged copies (SURVEY.md 8(d)); humanlike: "
                    "~45 %% of the bases in repeat families (SINE-, LINE-like, satellites, low complexity)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads timed through the CPU checker (rank 0, N=1), best of --cpu-reps; default 400 000 (5 000 with --ont: about 15 s of CPU work either way)")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--h2h-calls", type=int, default=3, help="whole-batch jnibwa_createAlignments calls (host request in, host response out) after the timed steps; the first one sizes buffers, the best of the rest is reported; 0 = skip")
    ap.add_argument("--paired", action="store_true", help="auxiliary measurement (BASELINE.json config 2): --reads is then the number of reads = 2 x pairs; "
                    "the library infers the insert-size statistics per call; metric/roofline fields are still reported for the seeding kernel")
    ap.add_argument("--ont", action="store_true", help="auxiliary measurement (BASELINE.json config 5): ONT-style error model (8 %% substitutions, 3 %% insertions, "
                    "3 %% deletions) for --read-len in the kilobases; use with a few thousand --reads")
    ap.add_argument("--pestat", default=None, help="with --paired: LOW,HIGH,AVG,STD of the FR insert size supplied by the caller (BwaMemAligner's "
                    "proper-pair statistics) instead of inferred per call; the call is then a single pass over the tiles")
    ap.add_argument("--keep-image", default=None, help="also copy the index image to this path (profiling helper)")
    ap.add_argument("--dump-request", default=None, help="write the first --cpu-sample reads as a request file (profiling helper)")
    ap.add_argument("--no-secondary", action="store_true", help="default single-GPU run only: skip the second measurement on the human-like genome (reported under `secondary.humanlike`)")
    ap.add_argument("--in-process-devices", default=os.environ.get("BENCH_INPROC_DEVICES", "auto"), help="N = 1 only: after the timed steps, one more index handle over these devices (BWAMEM_HIP_DEVICES syntax: "
                    "'all', '0,1,2,3', '0,0' = two replicas on one GPU for a rehearsal; 'auto' = all when more than one is visible; 'none') and the whole batch through jnibwa_createAlignments on it: what a single "
                    "JVM with one BwaMemIndex gets from a multi-GPU node; reported under `in_process_multi_device`")
    ap.add_argument("--dump-only", action="store_true", help="with --keep-image / --dump-request: stop once the files are written (profiling helper for the torch-free driver)")
    ap.add_argument("--image", default=os.environ.get("BWAHIP_REF_IMG"), help="use an existing .img (e.g. GATK's Homo_sapiens_assembly38.fasta.img) instead of "
                    "the synthetic genome; reads are sampled from its packed reference (default: $BWAHIP_REF_IMG)")
    args = ap.parse_args()
    if args.cpu_sample is None:
        args.cpu_sample = 5_000 if args.ont else 400_000
    return args


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher: start N ranks (one per GPU) as child processes BEFORE this process imports
    torch or touches HIP (a process that has initialised the GPU must never exec or fork GPU work), relay their output
    (rank 0 prints the JSON line) and fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # the contract is ONE JSON line on stdout: the ranks' stdout is captured and only rank 0's JSON line is passed on
        # (communication libraries print banners there); stderr goes through untouched
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, text=True))
    import threading
    outs = [""] * n

    def drain(i):
        outs[i] = procs[i].stdout.read()
    th = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(n)]
    for t in th:
        t.start()
    rc = 0
    while any(pr.poll() is None for pr in procs):              # a rank that dies leaves the others waiting in a barrier: end them
        bad = [r for r, pr in enumerate(procs) if pr.poll() not in (None, 0)]
        if bad:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
            break
        time.sleep(0.2)
    for t in th:
        t.join(timeout=30)
    for r, pr in enumerate(procs):
        c = pr.wait()
        if c != 0:
            print("[bench] rank %d exited with code %d" % (r, c), file=sys.stderr, flush=True)
            rc = rc or c or 1
    if not rc:
        lines = [l for l in outs[0].splitlines() if l.startswith("{")]
        if len(lines) != 1:
            print("[bench] rank 0 printed %d JSON lines" % len(lines), file=sys.stderr, flush=True)
            rc = 1
        else:
            print(lines[0], flush=True)
    sys.exit(rc)


def run_with_secondary(args):
    """The default single-GPU command: the BASELINE configuration (i.i.d. genome) as a child process, then the same batch size
    on the human-like genome as a second child, merged into one line -- `secondary.humanlike` is then evidence timed by whoever
    runs this command.  Children are started before this process imports torch or touches HIP; stderr passes through."""
    import subprocess
    def child(argv, timeout):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, stdout=subprocess.PIPE, text=True, timeout=timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return r.returncode, (json.loads(lines[-1]) if lines else None)
    rc, out = child(sys.argv[1:] + ["--no-secondary"], 3000)
    if rc != 0 or out is None:
        raise SystemExit(rc or 1)
    t0 = time.time()
    try:
        rc2, sec = child(["--genome", "humanlike", "--steps", "3", "--warmup", "2", "--h2h-calls", "2", "--cpu-sample", "100000", "--cpu-reps", "1",
                          "--no-secondary", "--in-process-devices", "none", "--reads", str(args.reads), "--read-len", str(args.read_len)], 1500)
        if rc2 == 0 and sec is not None:
            out["secondary"] = {"humanlike": {
                "what": "the same batch size on the synthetic human-like genome (~45 % of the bases in repeat families; bench.py --genome humanlike --steps 3 --warmup 2: the first calls on an index learn its tile and buffer sizes): "
                        "nearer to what a production hg38 run sees than the headline configuration",
                "workload": sec["config"]["workload"], "value": sec["value"], "unit": sec["unit"], "ms_per_step": sec["ms_per_step"], "steps": sec["steps"],
                "value_at_boundary": sec.get("value_at_boundary"), "host_to_host": sec.get("host_to_host"), "kernel_ms_isolated_pass": sec.get("kernel_ms_isolated_pass"),
                "roofline_frac": sec["roofline"]["frac"], "per_read": sec.get("per_read"), "cpu_baseline": sec.get("cpu_baseline"), "parity_sample": sec.get("parity_sample"),
                "parity_sample_tail": sec.get("parity_sample_tail"), "seconds": round(time.time() - t0, 1)}}
        else:
            out["secondary"] = {"humanlike": {"error": "child exited with %d" % rc2}}
    except Exception as ex:          # the headline line must not depend on the additive measurement
        out["secondary"] = {"humanlike": {"error": repr(ex)}}
    print(json.dumps(out), flush=True)
    raise SystemExit(0)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args.gpus)
    if (not args.no_secondary and "RANK" not in os.environ and args.gpus <= 1 and not (args.paired or args.ont or args.image or args.dump_only or args.pestat)
            and args.genome == "iid" and args.genome_bp >= 1_000_000_000):
        run_with_secondary(args)

    # libbwamem_hip.so asks the HIP runtime for eight hardware queues when it is the process's first HIP user (a JVM); here torch
    # initialises HIP first, so the same default is set before that happens (pipeline.cpp: hip_runtime_defaults)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import numpy as np
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print("[bench] --gpus %d but the launcher started %d ranks: reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- exercises the N > 1 code path (launcher, shared image,
    # barriers, max-over-ranks timing) on a one-GPU box; not a measurement
    rehearsal = bool(os.environ.get("BENCH_REHEARSAL"))
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = load_lib()
    lib.bwamem_hip_set_device(local)

    # ---- reference + index (not timed)
    t0 = time.time()

    def note(msg):
        if rank == 0:
            print("[bench] %6.1fs %s" % (time.time() - t0, msg), file=sys.stderr, flush=True)

    own_image = not args.image
    t_index_build = None
    if args.image:
        img = args.image
        genome_desc = "reference image %s" % os.path.basename(img)
    else:
        if args.genome == "humanlike":
            codes, contigs = synth_genome_humanlike(torch, dev, args.genome_bp, args.contigs, 0x5EED)
            genome_desc = "synthetic human-like GRCh38-scale genome (%d bp, %d contigs, ~45%% of the bases in repeat families)" % (args.genome_bp, args.contigs)
        else:
            codes, contigs = synth_genome(torch, dev, args.genome_bp, args.contigs, 0x5EED)
            genome_desc = "synthetic GRCh38-scale genome (%d bp, %d contigs, 5%% diverged repeats)" % (args.genome_bp, args.contigs)
        note("synthetic genome ready")
        # every rank holds the same genome (it samples its own reads from it); the index is built once, by rank 0, and the
        # image file is shared: one copy in the page cache of the node instead of one per rank
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        img = os.path.join(tmpdir, "bwamem_hip_bench_%s.img" % (os.environ.get("MASTER_PORT", "p") + "_" + str(os.getppid()) if world > 1 else str(os.getpid())))
        if rank == 0:
            if os.environ.get("BENCH_TORCH_INDEX_BUILDER"):          # the independent torch builder (tests compare the two images)
                import index_build_gpu as G
                pieces = G.build_pieces(codes)
                note("suffix array / BWT / occ / SA built on the device (torch builder)")
                G.write_image(img, pieces, contigs)
                del pieces
            else:                                                     # the library's own builder (csrc/k_index.hip), codes handed over in host memory
                h_codes = codes.cpu().numpy()
                torch.cuda.empty_cache()
                names = (ctypes.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
                lens = (ctypes.c_int64 * len(contigs))(*[l for _, l in contigs])
                tb = time.time()
                if lib.bwamem_hip_build_image(h_codes.ctypes.data, int(h_codes.size), len(contigs), names, lens, img.encode()) != 0:
                    raise SystemExit("index build failed")
                t_index_build = time.time() - tb
                note("index image built by the library's device builder in %.1f s" % t_index_build)
                del h_codes
            torch.cuda.empty_cache()
            note("index image written (%.2f GB)" % (os.path.getsize(img) / 1e9))
    if dist is not None:
        dist.barrier()
    t_index = time.time() - t0
    idx = lib.jnibwa_openIndex(os.open(img, os.O_RDONLY))
    note("index resident in HBM")
    if not idx:
        raise SystemExit("openIndex failed")
    if dist is not None:
        dist.barrier()                       # every rank has the image mapped: rank 0 may unlink it at the end
    if args.image:                           # reads are sampled from the image's own packed reference
        codes, contigs = genome_from_index(torch, dev, lib, idx)
        args.genome_bp = int(codes.numel())
        args.contigs = len(contigs)
        note("packed reference of the image unpacked on the device (%d bp, %d contigs)" % (args.genome_bp, args.contigs))

    # ---- request resident in HBM (not timed)
    L, R = args.read_len, args.reads
    payload = (synth_pairs(torch, dev, codes, contigs, R // 2, L, 42 + rank) if args.paired else
               synth_long_reads(torch, dev, codes, contigs, R, L, 42 + rank) if args.ont else synth_reads(torch, dev, codes, contigs, R, L, 42 + rank))
    note("%d reads generated on the device" % R)
    del codes
    torch.cuda.empty_cache()
    if args.keep_image:
        import shutil
        shutil.copyfile(img, args.keep_image)
    if args.dump_request:
        S0 = min(max(args.cpu_sample, 1), R)
        with open(args.dump_request, "wb") as f:
            f.write(struct.pack("<i", S0)); f.write(payload[:S0].cpu().numpy().tobytes())
    if args.dump_only:
        lib.jnibwa_destroyIndex(idx)
        if own_image and rank == 0:
            os.unlink(img)
        return
    h_off = (np.arange(R + 1, dtype=np.int64) * (L + 1))
    torch.cuda.synchronize()
    batch = lib.bwamem_hip_batch_wrap_device(idx, payload.data_ptr(), R * (L + 1), R, h_off.ctypes.data)
    if not batch:
        raise SystemExit("batch_wrap_device failed")
    p = lib.jnibwa_createDefaultOptions()
    opts = ctypes.create_string_buffer(ctypes.string_at(p, 168), 168)
    lib.jnibwa_free(p)
    if args.paired:
        struct.pack_into("<i", opts, 60, struct.unpack_from("<i", opts, 60)[0] | 0x2 | int(os.environ.get("BENCH_EXTRA_FLAG", "0"), 0))   # mem_opt_t.flag |= MEM_F_PE

    pes = None
    if args.paired and args.pestat:
        lo, hi, avg, std = args.pestat.split(",")
        pes = b"".join(struct.pack("<iiiidd", int(lo), int(hi), 0, 0, float(avg), float(std)) if i == 1 else struct.pack("<iiiidd", 0, 0, 1, 0, 0.0, 0.0) for i in range(4))

    def step():
        if lib.bwamem_hip_batch_align(idx, opts, pes, batch, rank * R) != 0:
            raise SystemExit("align failed")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_times(x):
        """every rank's value of x -> list on every rank (the only torch.distributed traffic besides the barriers)"""
        if dist is None:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    for _ in range(args.warmup):
        step()
    note("%d warm-up steps done" % args.warmup)
    lib.bwamem_hip_stats_enable(1)
    lib.bwamem_hip_stats_reset()
    barrier()
    t1 = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_own = time.time() - t1
    per_rank = gather_times(elapsed_own)
    elapsed = max(per_rank)
    st_timed = Stats(); lib.bwamem_hip_stats_get(ctypes.byref(st_timed))
    lib.bwamem_hip_stats_enable(0)
    result_bytes = lib.bwamem_hip_batch_result_bytes(batch)
    note("%d timed steps done (%.1f ms per step)" % (args.steps, elapsed / args.steps * 1e3))

    # ---- the same batch through the drop-in entry point itself: jnibwa_createAlignments, host request in, malloc'ed host
    # response out (jnibwa.c:197-235) -- every rank at once, so that at N > 1 the ranks share the host's memory and PCIe
    # like N GATK executors would.  Reported next to `value`, never as `value` (which is quoted with the request resident).
    h2h = None
    inproc_spec = args.in_process_devices
    if inproc_spec == "auto":
        inproc_spec = "all" if lib.bwamem_hip_device_count() > 1 else "none"
    want_inproc = rank == 0 and world == 1 and inproc_spec != "none" and args.h2h_calls > 0
    keep_full = keep_resident = None
    if args.h2h_calls > 0:
        full = np.empty(4 + R * (L + 1), dtype=np.uint8)
        full[:4] = np.frombuffer(struct.pack("<i", R), dtype=np.uint8)
        full[4:] = payload.reshape(-1).cpu().numpy()
        resident = None
        if rank == 0:
            resident = np.empty(max(result_bytes, 1), dtype=np.uint8)
            if lib.bwamem_hip_batch_download(batch, resident.ctypes.data) != 0:
                raise SystemExit("download failed")
        secs, same, h2h_bytes = [], None, 0
        for k in range(args.h2h_calls):
            sz = ctypes.c_size_t()
            barrier()
            tj = time.time()
            gp = lib.jnibwa_createAlignments(idx, opts, pes, full.ctypes.data, ctypes.byref(sz))
            dt = time.time() - tj
            if not gp:
                raise SystemExit("jnibwa_createAlignments failed")
            secs.append(max(gather_times(dt)))
            h2h_bytes = sz.value
            if rank == 0 and k == args.h2h_calls - 1 and (not args.paired or pes is not None or True):
                got = np.ctypeslib.as_array(ctypes.cast(gp, ctypes.POINTER(ctypes.c_ubyte)), shape=(max(sz.value, 1),))
                same = bool(sz.value == result_bytes and np.array_equal(got[:sz.value], resident[:result_bytes]))
            lib.jnibwa_free(gp)
            note("jnibwa_createAlignments call %d of %d: %.3f s" % (k + 1, args.h2h_calls, secs[-1]))
        best = min(secs[1:]) if len(secs) > 1 else secs[0]
        h2h = {"reads_per_s": world * R / best, "seconds_per_call": [round(x, 4) for x in secs], "calls": args.h2h_calls,
               "request_bytes_per_gpu": int(full.nbytes), "response_bytes_per_gpu": int(h2h_bytes), "identical_to_resident_response": same,
               "what": "jnibwa_createAlignments on the whole batch: pageable host request in, malloc'ed host response out, PCIe both ways; "
                       "max over ranks per call, first call sizes buffers, best of the rest"}
        if want_inproc:
            keep_full, keep_resident = full, resident
        del full, resident

    # keep the records of the LAST reads of the timed batch (four tiles and a seeding chunk in flight) for the parity check below
    tail_bytes, S2 = None, 0
    if rank == 0 and world == 1 and args.cpu_sample > 0 and (not args.paired or pes is not None):   # (paired with inferred statistics: the batch-wide ones differ from a slice's)
        import bwalib as B
        if not os.path.exists(B.ORACLE_LIB):
            B.build_oracle()
        orc_dll = ctypes.CDLL(B.ORACLE_LIB)
        S2 = min(args.cpu_sample, 50000, R)
        total = lib.bwamem_hip_batch_result_bytes(batch)
        buf = (ctypes.c_ubyte * total)()
        if lib.bwamem_hip_batch_download(batch, buf) == 0 and hasattr(orc_dll, "oracle_response_offsets"):
            offs = (ctypes.c_int64 * (R + 1))()                 # record boundaries of the whole response (C walk in the checker library)
            ro = orc_dll.oracle_response_offsets
            ro.restype = ctypes.c_int
            ro.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p]
            if ro(buf, total, R, offs) != 0 or offs[R] != total:
                raise SystemExit("response of the timed batch does not parse")
            tail_bytes = bytes(memoryview(buf)[offs[R - S2]:offs[R]])
        del buf
    # Roofline pass (untimed): during the timed steps several tiles are in flight on separate streams, so the
    # HIP-event interval of one launch also contains other tiles' kernels.  One extra step with a single tile in
    # flight gives per-launch durations that mean what rocprofv3 --kernel-trace reports for the same kernel.
    streams_env = os.environ.get("BWAMEM_HIP_STREAMS")
    os.environ["BWAMEM_HIP_STREAMS"] = "1"
    os.environ["BWAMEM_HIP_SEED_AHEAD"] = "0"            # seeding chunks strictly between the tiles, nothing else on the GPU
    lib.bwamem_hip_stats_enable(1)
    lib.bwamem_hip_stats_reset()
    step()
    torch.cuda.synchronize()
    st = Stats(); lib.bwamem_hip_stats_get(ctypes.byref(st))
    note("roofline pass (one tile in flight) done")
    del os.environ["BWAMEM_HIP_SEED_AHEAD"]
    if streams_env is None:
        del os.environ["BWAMEM_HIP_STREAMS"]
    else:
        os.environ["BWAMEM_HIP_STREAMS"] = streams_env
    lib.bwamem_hip_stats_enable(0)

    out = None
    if rank == 0:
        value = world * R * args.steps / elapsed
        kern = dict(encode=st.ms_encode, seed=st.ms_seed, sa=st.ms_sa, chain=st.ms_chain, extend=st.ms_extend, post=st.ms_post, final=st.ms_final, pack=st.ms_pack, other=st.ms_other)
        # roofline of the occurrence-table gather kernel (k_seed): 2 x 32-byte occ blocks (device layout) per interval extension
        n_launch = max(1, st.n_launch_seed)
        alg_bytes = 64.0 * st.n_ext / n_launch
        avg_s = st.ms_seed * 1e-3 / n_launch
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        # HBM traffic of the same kernel: PMC counters cannot be read from inside this process, so the per-extension
        # FETCH_SIZE measured by tests/gpu_units/pmc_seed.sh on this workload (committed summary, see profiles/README.md)
        # is scaled to this launch's extension count
        traffic, traffic_note = None, None
        for name in ("r03_pmc_k_seed.json", "r02_pmc_k_seed.json", "r01_pmc_k_seed.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                traffic = pmc["fetch_bytes_per_ext"] * st.n_ext / n_launch
                traffic_note = pmc.get("note")
                break
            except (OSError, KeyError, ValueError):
                pass
        nr = max(1, st.n_reads)
        out = {
            "metric": "150bp reads aligned/sec vs GRCh38-scale reference (1/2/4/8 MI355X)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32/int64", "data": "synthetic",
            "config": {"workload": "%d x %dbp %s synthetic reads per GPU vs %s, full index in HBM" % (R, L, "paired-end (2 x %d pairs)" % (R // 2) if args.paired else "single-end ONT-style (8 %% sub, 3 %% ins, 3 %% del)" if args.ont else "single-end", genome_desc),
                       "reads_per_gpu": R, "read_len": L, "genome_bp": args.genome_bp, "genome": "image" if args.image else args.genome, "index_build_s": round(t_index, 1), "index_builder_s": round(t_index_build, 1) if t_index_build is not None else None, "response_bytes": result_bytes,
                       "parallelism": "read-sharded x%d, no collectives" % world, "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "paired_end": bool(args.paired), "insert_size_statistics": ("supplied " + args.pestat) if (args.paired and args.pestat) else ("inferred per call" if args.paired else None),
                       "value_is": "device-resident rate: request already in HBM when the timed region starts, response left in HBM (the bench contract); the rate through jnibwa_createAlignments itself is `host_to_host`"},
            "per_rank": {"reads_per_s": [R * args.steps / x for x in per_rank], "ms_per_step_min": min(per_rank) / args.steps * 1e3, "ms_per_step_max": max(per_rank) / args.steps * 1e3},
            "value_at_boundary": h2h["reads_per_s"] if h2h else None,
            "value_at_boundary_is": "reads/s through jnibwa_createAlignments itself (SURVEY.md 8(d): wall inside the call, H2D and D2H included): what a GATK caller gets; details in `host_to_host`",
            "host_to_host": h2h,
            "roofline": {"bound": "hbm", "kernel": "k_seed", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_note, "measured_in": "extra untimed step with nothing else on the GPU: one tile in flight, seeding chunks not overlapped (see DESIGN.md section 5)", "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_s * 1e3, "launches": int(st.n_launch_seed),
                         "n_ext_per_read": st.n_ext / nr},
            "kernel_ms_isolated_pass": {k: round(v, 2) for k, v in kern.items()},
            "tiles_in_flight_timed": int(os.environ.get("BWAMEM_HIP_STREAMS", "4")),
            "counters": {"n_ext": int(st.n_ext), "n_lf": int(st.n_lf), "n_sa": int(st.n_sa), "n_dp_cells": int(st.n_dp_cells), "tiles": int(st.n_tiles), "retries": int(st.n_retries)},
            "per_read": {"ext": st.n_ext / nr, "sa_lookups": st.n_sa / nr, "dp_cells": st.n_dp_cells / nr, "response_bytes": result_bytes / max(1, R)},
        }

    # ---- CPU baseline + parity sample (rank 0, N = 1 only)
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import bwalib as B
        if not os.path.exists(B.ORACLE_LIB):
            B.build_oracle()
        orc, kind, kind_note = B.oracle_lib(), "port", "oracle/ (own CPU restatement, not libbwa)"
        stock = B.stock_libbwa()               # LIBBWA_PATH / BWA_ORACLE_SRC: a stock libbwa behind the same jnibwa_* ABI, when the run environment has one
        if stock is not None:
            orc, kind, kind_note = stock, "reference", "stock libbwa (%s)" % stock.path
        S = min(args.cpu_sample, R)
        req = struct.pack("<i", S) + payload[:S].cpu().numpy().tobytes()
        # the GPU box hands one GPU's share of the host to this process (16 cores); stay inside it
        cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        ho = orc.open_index(img)
        oo = B.set_opt(orc.default_options(), n_threads=cores)
        if args.paired:
            B.set_opt(oo, flag=B.get_opt(oo, "flag") | B.MEM_F_PE)
        tcs = []
        for _ in range(max(1, args.cpu_reps)):
            tc = time.time()
            want = orc.align_raw(ho, oo, req, pes)
            tcs.append(time.time() - tc)
            note("CPU checker: %d reads in %.1f s" % (S, tcs[-1]))
        tcpu = min(tcs)
        orc.destroy_index(ho)
        rb = ctypes.create_string_buffer(req, len(req)); sz = ctypes.c_size_t()
        gp = lib.jnibwa_createAlignments(idx, opts, pes, rb, ctypes.byref(sz))
        got = ctypes.string_at(gp, sz.value) if gp else None
        if gp:
            lib.jnibwa_free(gp)
        ident = None
        if got is not None:
            a, b = B.split_response(got, S), B.split_response(want, S)
            ident = sum(1 for x, y in zip(a, b) if x == y) / S
        # second sample: the LAST reads of the batch, taken from the big batch's own response (so the seeding chunks, tiles
        # and interval-store reuse of a 10 M-read call are what is being checked), against the checker run on that slice
        # with its position in the call (the tie-breaking hash of mem_mark_primary_se uses the read index)
        tail = None
        if tail_bytes is not None and kind == "port" and hasattr(orc.dll, "oracle_createAlignmentsAt"):
            req2 = struct.pack("<i", S2) + payload[R - S2:].cpu().numpy().tobytes()
            fn = orc.dll.oracle_createAlignmentsAt
            fn.restype = ctypes.c_void_p
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int64]
            ho2 = orc.open_index(img)
            rb2 = ctypes.create_string_buffer(req2, len(req2)); sz2 = ctypes.c_size_t()
            ob2 = ctypes.create_string_buffer(bytes(oo), 168)
            p2 = fn(ho2, ob2, pes, rb2, ctypes.byref(sz2), rank * R + R - S2)
            want2 = ctypes.string_at(p2, sz2.value) if p2 else None
            orc.destroy_index(ho2)
            if want2 is not None:
                a2, b2 = B.split_response(tail_bytes, S2), B.split_response(want2, S2)
                tail = {"reads": S2, "from": "the last reads of the timed batch's own response", "frac_identical_records": sum(1 for x, y in zip(a2, b2) if x == y) / S2}
        out["cpu_baseline"] = {"value": S / tcpu, "unit": "reads/s", "cores": cores, "kind": kind,
                               "sample": "first %d reads of the same batch through %s, %d threads, best of %d runs (%s s)" % (S, kind_note, cores, len(tcs), ", ".join("%.1f" % x for x in tcs))}
        out["parity_sample"] = {"reads": S, "frac_identical_records": ident, "checker": kind_note}
        if tail is not None:
            out["parity_sample_tail"] = tail

    lib.bwamem_hip_batch_free(batch)
    lib.jnibwa_destroyIndex(idx)
    if want_inproc and keep_full is not None:
        # One process, one index handle, several devices (pipeline.cpp: jnibwa_openIndex / create_alignments_split): the first
        # handle is gone, so device 0 holds one replica like the others.  Additive: a failure here is recorded, not fatal.
        try:
            del payload
            torch.cuda.empty_cache()
            os.environ["BWAMEM_HIP_DEVICES"] = inproc_spec
            tl = time.time()
            idx2 = lib.jnibwa_openIndex(os.open(img, os.O_RDONLY))
            del os.environ["BWAMEM_HIP_DEVICES"]
            if not idx2:
                raise RuntimeError("openIndex over %s failed" % inproc_spec)
            nrep = lib.bwamem_hip_index_replicas(idx2)
            note("index handle over %d replicas (%s) open after %.1f s" % (nrep, inproc_spec, time.time() - tl))
            secs2, same2 = [], None
            for k in range(3):
                sz = ctypes.c_size_t()
                tj = time.time()
                gp = lib.jnibwa_createAlignments(idx2, opts, pes, keep_full.ctypes.data, ctypes.byref(sz))
                secs2.append(time.time() - tj)
                if not gp:
                    raise RuntimeError("jnibwa_createAlignments on the multi-device handle failed")
                if k == 2:
                    got = np.ctypeslib.as_array(ctypes.cast(gp, ctypes.POINTER(ctypes.c_ubyte)), shape=(max(sz.value, 1),))
                    same2 = bool(sz.value == result_bytes and np.array_equal(got[:sz.value], keep_resident[:result_bytes]))
                lib.jnibwa_free(gp)
                note("multi-device jnibwa_createAlignments call %d of 3: %.3f s" % (k + 1, secs2[-1]))
            lib.jnibwa_destroyIndex(idx2)
            out["in_process_multi_device"] = {"devices": inproc_spec, "replicas": int(nrep), "reads_per_s": R / min(secs2[1:]), "seconds_per_call": [round(x, 4) for x in secs2],
                                              "identical_to_single_device_response": same2, "index_open_s": round(time.time() - tl - sum(secs2), 1),
                                              "what": "one process, one jnibwa_openIndex handle with a replica per device, the whole batch as ONE jnibwa_createAlignments call cut across the replicas "
                                                      "(host walk of the length-less request, per-device streaming, responses concatenated); best of calls 2-3"}
        except Exception as ex:
            out["in_process_multi_device"] = {"devices": inproc_spec, "error": repr(ex)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if rank == 0 and own_image:              # (the other ranks mapped it before the barrier above; an unlinked file lives until unmapped)
        try:
            os.unlink(img)
        except OSError:
            pass
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
