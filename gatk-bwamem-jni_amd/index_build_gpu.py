"""Device-side index builder: 2-bit reference -> suffix array of forward+reverse-complement
-> BWT with interleaved occ checkpoints + sampled SA -> the ".img" layout the library opens.

Covers the "next" row (f)1 of SURVEY.md section 8 (bwa_idx_build replacement,
...BwaMemIndex.c:42-63) at GRCh38 scale, where the host builder in csrc/index_build.cpp
(prefix-doubling on std::sort) is far too slow.  Suffix sorting is an MSD bucket sort on
32-mer keys: suffixes are bucketed by their first two bases, each bucket is radix-sorted on the
next 32 bases (one 64-bit key), and only groups that are still tied are refined with further
32-mers -- on a random-like genome almost every suffix is resolved by the first key, and the
planted repeats need a few dozen rounds on a quickly shrinking set.  PyTorch supplies device
memory and the sort/scan primitives; output is byte-identical to the host builder
(tests/test_index_gpu_builder.py).
"""
import struct

import numpy as np
import torch

SIGN = -(1 << 63)
CHUNK = 1 << 30          # several torch ops (nonzero, flip) are limited to < 2^31 elements per call


def _nonzero_chunked(mask_fn, n, device):
    """indices i in [0, n) with mask_fn(lo, hi)[i - lo] true, evaluated CHUNK elements at a time"""
    out = []
    for lo in range(0, n, CHUNK):
        hi = min(n, lo + CHUNK)
        idx = torch.nonzero(mask_fn(lo, hi)).squeeze(1)
        if idx.numel():
            out.append(idx + lo)
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.int64, device=device)


def _revcomp(fwd):
    n = fwd.numel()
    out = torch.empty_like(fwd)
    for lo in range(0, n, CHUNK):
        hi = min(n, lo + CHUNK)
        out[n - hi: n - lo] = 3 - torch.flip(fwd[lo:hi], [0])
    return out


def _pack_words(T, n_words, chunk=1 << 26):
    """W[k] = bases 32k..32k+31, first base in the two most significant bits (zero padded)."""
    dev = T.device
    n = T.numel()
    W = torch.zeros(n_words, dtype=torch.int64, device=dev)
    shifts = (62 - 2 * torch.arange(32, device=dev, dtype=torch.int64))
    for w0 in range(0, (n + 31) // 32, chunk):
        w1 = min((n + 31) // 32, w0 + chunk)
        seg = T[w0 * 32: min(n, w1 * 32)].to(torch.int64)
        if seg.numel() < (w1 - w0) * 32:
            seg = torch.cat([seg, torch.zeros((w1 - w0) * 32 - seg.numel(), dtype=torch.int64, device=dev)])
        W[w0:w1] = (seg.view(-1, 32) << shifts).sum(1)
    return W


def _u32_bits(x):
    """int64 values in [0, 2^32) -> int32 tensor with the same 32 bits"""
    return torch.where(x >= (1 << 31), x - (1 << 32), x).to(torch.int32)


def _key32(W, p):
    """unsigned-order-preserving int64 key of the 32-mer starting at text position p (beyond the end = A...)."""
    idx = torch.clamp(p >> 5, max=W.numel() - 2)
    s = (p & 31) << 1
    a = W[idx]
    b = W[idx + 1]
    hi = a << s
    sh = torch.clamp(64 - s, max=63)
    lo = (b >> sh) & ((torch.ones_like(s) << s) - 1)
    lo = torch.where(s == 0, torch.zeros_like(lo), lo)
    return (hi | lo) ^ SIGN


def _sort_bucket(W, pos, n, first_off):
    """positions of one bucket -> the same positions in suffix order."""
    key = _key32(W, pos + first_off)
    key, perm = torch.sort(key, stable=True)
    order = pos[perm]
    del perm
    m = order.numel()
    if m < 2:
        return order
    same = key[1:] == key[:-1]
    del key
    tied = torch.zeros(m, dtype=torch.bool, device=pos.device)
    tied[1:] |= same
    tied[:-1] |= same
    slots = torch.nonzero(tied).squeeze(1)
    if slots.numel() == 0:
        return order
    # group id = slot of the group's first element
    starts = torch.ones(m, dtype=torch.bool, device=pos.device)
    starts[1:] = ~same
    del same, tied
    gid_all = torch.cummax(torch.where(starts, torch.arange(m, device=pos.device), torch.zeros(1, dtype=torch.int64, device=pos.device)), 0).values
    grp = gid_all[slots]
    del gid_all, starts
    p = order[slots]
    off = first_off + 32
    while slots.numel() > 0:
        beyond = (p + off) >= n
        k = _key32(W, torch.clamp(p + off, max=n))
        # beyond-the-end suffixes are all-zero from here on; among themselves the shorter (larger pos) sorts first
        k = torch.where(beyond, torch.full_like(k, SIGN), k)
        sec = torch.where(beyond, -p, torch.zeros_like(p))
        # sort by (grp, key, sec): three stable passes, least significant first
        o = torch.sort(sec, stable=True).indices
        o = o[torch.sort(k[o], stable=True).indices]
        o = o[torch.sort(grp[o], stable=True).indices]
        p, k, grp, beyond, sec = p[o], k[o], grp[o], beyond[o], sec[o]
        order[slots] = p                       # slots stay ascending; groups stay inside their slot ranges
        eq = (grp[1:] == grp[:-1]) & (k[1:] == k[:-1]) & ~(beyond[1:] & beyond[:-1])
        t = torch.zeros(p.numel(), dtype=torch.bool, device=pos.device)
        t[1:] |= eq
        t[:-1] |= eq
        if not bool(t.any()):
            break
        st = torch.ones(p.numel(), dtype=torch.bool, device=pos.device)
        st[1:] = ~eq
        newg = torch.cummax(torch.where(st, slots, torch.zeros(1, dtype=torch.int64, device=pos.device)), 0).values
        keep = torch.nonzero(t).squeeze(1)
        slots, p, grp = slots[keep], p[keep], newg[keep]
        off += 32
    return order


def suffix_array(T):
    """T: uint8 codes 0..3 of forward+reverse-complement.  Returns int64 SA (implicit smallest sentinel at n)."""
    dev = T.device
    n = T.numel()
    W = _pack_words(T, (n + 31) // 32 + 4)
    code = torch.empty_like(T)
    for lo in range(0, n, CHUNK):
        hi = min(n, lo + CHUNK)
        nxt = torch.zeros(hi - lo, dtype=torch.uint8, device=dev)
        m = min(n, hi + 1) - (lo + 1)
        nxt[:m] = T[lo + 1: lo + 1 + m]
        code[lo:hi] = T[lo:hi] * 4 + nxt
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    at = 0
    for b in range(16):
        pos = _nonzero_chunked(lambda lo, hi: code[lo:hi] == b, n, dev)
        if pos.numel():
            sa[at: at + pos.numel()] = _sort_bucket(W, pos, n, 2)
            at += pos.numel()
        del pos
    del code
    return sa


def build_pieces(fwd):
    """fwd: uint8 tensor of base codes (no ambiguous bases left).  Returns dict of numpy pieces."""
    dev = fwd.device
    l_pac = fwd.numel()
    T = torch.empty(2 * l_pac, dtype=torch.uint8, device=dev)
    T[:l_pac] = fwd
    T[l_pac:] = _revcomp(fwd)
    n = T.numel()
    sa = suffix_array(T)
    # sentinel-inclusive ranks: rank 0 = empty suffix; rank k = sa[k-1]
    zero_at = int(_nonzero_chunked(lambda lo, hi: sa[lo:hi] == 0, n, dev)[0])
    primary = zero_at + 1
    # BWT without the sentinel position: ranks 0..n with rank `primary` left out
    step = 1 << 28
    Bfull = torch.empty(n + 1, dtype=torch.uint8, device=dev)
    Bfull[0] = T[n - 1]                                    # rank 0: the symbol before the sentinel
    for i in range(0, n, step):
        j = min(n, i + step)
        Bfull[1 + i: 1 + j] = T[torch.clamp(sa[i:j] - 1, min=0)]
    B = torch.empty(n, dtype=torch.uint8, device=dev)
    B[:primary] = Bfull[:primary]
    B[primary:] = Bfull[primary + 1:]
    del Bfull
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    for i in range(0, n, step):
        counts += torch.bincount(T[i: i + step].to(torch.int64), minlength=4)[:4]
    L2 = [0]
    for c in range(4):
        L2.append(L2[-1] + int(counts[c]))
    # interleaved occ + bwt words
    nblk = (n + 127) // 128
    pad = nblk * 128 - n
    if pad:
        Bp = torch.zeros(nblk * 128, dtype=torch.uint8, device=dev)
        Bp[:n] = B
    else:
        Bp = B
    blk = torch.zeros((nblk, 16), dtype=torch.int32, device=dev)
    tot = []
    for c in range(4):
        per = torch.zeros(nblk, dtype=torch.int64, device=dev)
        for i in range(0, nblk, 1 << 21):
            j = min(nblk, i + (1 << 21))
            seg = (Bp[i * 128: j * 128] == c).view(-1, 128).sum(1)
            per[i:j] = seg
        if pad and c == 0:
            per[-1] -= pad                                 # the padding was counted as 'A'
        excl = torch.cumsum(per, 0) - per
        tot.append(int(per.sum()))
        blk[:, 2 * c] = _u32_bits(excl & 0xffffffff)
        blk[:, 2 * c + 1] = _u32_bits(excl >> 32)
    shifts = (30 - 2 * torch.arange(16, device=dev, dtype=torch.int64))
    for i in range(0, nblk, 1 << 21):
        j = min(nblk, i + (1 << 21))
        wv = (Bp[i * 128: j * 128].to(torch.int64).view(-1, 16) << shifts).sum(1)          # values < 2^32
        blk[i:j, 8:] = _u32_bits(wv).view(-1, 8)
    flat = blk.view(-1)
    n_words = (n + 15) // 16 + 8 * nblk                     # words before the final count record
    last = torch.tensor([v for c in range(4) for v in (tot[c] & 0xffffffff, tot[c] >> 32)], dtype=torch.int64, device=dev)
    last = _u32_bits(last)
    bwt = torch.cat([flat[:n_words], last])
    # sampled SA
    n_sa = (n + 32) // 32
    ranks = torch.arange(1, n_sa, device=dev, dtype=torch.int64) * 32 - 1
    sa_s = torch.cat([torch.tensor([-1], dtype=torch.int64, device=dev), sa[ranks]])
    # packed forward strand
    npac = l_pac // 4 + 1
    fp = torch.cat([fwd, torch.zeros(npac * 4 - l_pac, dtype=torch.uint8, device=dev)]).view(-1, 4).to(torch.int32)
    pac = ((fp[:, 0] << 6) | (fp[:, 1] << 4) | (fp[:, 2] << 2) | fp[:, 3]).to(torch.uint8)
    return dict(primary=primary, L2=L2, seq_len=n, bwt=bwt.cpu().numpy().view(np.uint32), sa=sa_s.cpu().numpy().view(np.uint64),
                l_pac=l_pac, pac=pac.cpu().numpy())


def write_image(path, pieces, contigs, seed=11):
    """contigs: list of (name, length).  Layout: SURVEY.md App. A.4 (pointer fields written as zero)."""
    hdr = bytearray(1120)
    struct.pack_into("<Q5QQQ", hdr, 0, pieces["primary"], *pieces["L2"], pieces["seq_len"], pieces["bwt"].size)
    for i in range(256):
        x = 0
        for j in range(4):
            x |= (((i & 3) == j) + ((i >> 2 & 3) == j) + ((i >> 4 & 3) == j) + ((i >> 6) == j)) << (j << 3)
        struct.pack_into("<I", hdr, 72 + 4 * i, x)
    struct.pack_into("<i", hdr, 1096, 32)
    struct.pack_into("<Q", hdr, 1104, pieces["sa"].size)
    bns = bytearray(48)
    struct.pack_into("<qiI", bns, 0, pieces["l_pac"], len(contigs), seed)
    anns, strs, off = b"", b"", 0
    for name, ln in contigs:
        a = bytearray(40)
        struct.pack_into("<qiiIi", a, 0, off, ln, 0, 0, 0)
        anns += bytes(a)
        strs += name.encode() + b"\0" + b"\0"
        off += ln
    with open(path, "wb") as f:
        f.write(hdr)
        pieces["bwt"].tofile(f)
        pieces["sa"].tofile(f)
        f.write(bns); f.write(anns); f.write(strs)
        pieces["pac"].tofile(f)
