"""Function-level parity: single device functions of the product (wrapped by tests/gpu_units/units.hip)
against the oracle's counterparts on random inputs.  The emulation flavour runs in the CPU suite;
the hipcc flavour is the GPU test (it is what caught the hipcc -O3 introsort miscompile)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import bwalib as B

UNITS = os.path.join(B.ROOT, "tests", "gpu_units")


def _load(flavour):
    subprocess.run(["make", "-s", "-C", UNITS] + (["emu"] if flavour == "emu" else []), check=True)
    lib = ctypes.CDLL(os.path.join(UNITS, "_build", "libunits_%s.so" % flavour))
    lib.unit_sort_pairs.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.unit_extend.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    return lib


def _sort_cases(rng, sizes, n_keys):
    for n in sizes:
        for k in n_keys:
            x = rng.integers(0, k, size=n, dtype=np.uint64)
            yield np.stack([x, np.arange(n, dtype=np.uint64)], axis=1).copy()
    # adversarial shapes: sorted, reversed, all equal, organ pipe
    for n in (3, 17, 40, 200):
        a = np.arange(n, dtype=np.uint64)
        for x in (a, a[::-1].copy(), np.zeros(n, dtype=np.uint64), np.minimum(a, a[::-1])):
            yield np.stack([x, np.arange(n, dtype=np.uint64)], axis=1).copy()


def _check_sort(units, oracle, sizes):
    osort = oracle.dll.oracle_test_sort_pairs
    osort.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    rng = np.random.default_rng(3)
    for xy in _sort_cases(rng, sizes, (2, 5, 1000)):
        want = xy.copy(); osort(len(want), want.ctypes.data)
        got = xy.copy(); assert units.unit_sort_pairs(len(got), got.ctypes.data) == 0
        assert (got == want).all(), "tie permutation differs for n=%d" % len(xy)


def _check_extend(units, oracle, n_cases, max_q):
    oext = oracle.dll.o_ksw_extend2
    oext.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(11)
    base_opts = oracle.default_options()
    for it in range(n_cases):
        qlen = int(rng.integers(1, max_q)); tlen = int(rng.integers(1, max_q + 60))
        t = rng.integers(0, 4, size=tlen, dtype=np.uint8)
        # query = mutated copy of the target so that the DP actually extends
        q = np.resize(t, qlen).copy()
        mut = rng.random(qlen) < rng.choice([0.0, 0.01, 0.02, 0.1, 0.4])
        q[mut] = rng.integers(0, 5, size=int(mut.sum()), dtype=np.uint8)
        if rng.random() < 0.4 and qlen > 12:                       # an indel
            cut = int(rng.integers(3, qlen - 3)); k = int(rng.integers(1, 8))
            q = np.concatenate([q[:cut], q[cut + k:]]) if rng.random() < 0.5 else np.concatenate([q[:cut], rng.integers(0, 4, size=k, dtype=np.uint8), q[cut:]])
            qlen = len(q)
        kw = dict(a=int(rng.choice([1, 2])), b=int(rng.choice([4, 9, 2])), o_del=int(rng.choice([6, 16, 0])), e_del=int(rng.choice([1, 2])),
                  o_ins=int(rng.choice([6, 16, 0])), e_ins=int(rng.choice([1, 3])))
        w = int(rng.choice([100, 200, 5, 1])); zdrop = int(rng.choice([100, 0, 10])); end_bonus = int(rng.choice([5, 0])); h0 = int(rng.integers(0, 150))
        if it % 2 == 0:
            # the shape production sees most: the extension of a long seed of a well-placed read -- default penalties, the target
            # longer than the query, a handful of substitutions, now and then an N or a short indel
            kw = dict(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1)
            if rng.random() < 0.3:
                kw.update(b=int(rng.choice([4, 9])), o_del=int(rng.choice([6, 16])), o_ins=int(rng.choice([6, 16])), e_del=int(rng.choice([1, 2])))
            qlen = int(rng.integers(2, min(max_q, 200))); tlen = qlen + int(rng.integers(0, 90))
            t = rng.integers(0, 4, size=tlen, dtype=np.uint8)
            q = t[:qlen].copy()
            for pos in rng.integers(0, qlen, size=int(rng.choice([0, 1, 1, 2, 2, 3, 4]))):
                q[pos] = (q[pos] + int(rng.integers(1, 4))) % 4
            if rng.random() < 0.2:
                q[int(rng.integers(0, qlen))] = 4
            if rng.random() < 0.15 and qlen > 12:
                cut = int(rng.integers(3, qlen - 3)); k = int(rng.integers(1, 4))
                q = np.concatenate([q[:cut], q[cut + k:]]) if rng.random() < 0.5 else np.concatenate([q[:cut], rng.integers(0, 4, size=k, dtype=np.uint8), q[cut:]])
                qlen = len(q)
            h0 = int(rng.integers(19, 130)); w = int(rng.choice([100, 100, 200, 9])); zdrop = int(rng.choice([100, 100, 10])); end_bonus = 5
        opts = B.set_opt(bytearray(base_opts), **kw)
        mat = []
        for i in range(4):
            mat += [kw["a"] if i == j else -kw["b"] for j in range(4)] + [-1]
        mat += [-1] * 5
        B.set_opt(opts, mat=mat)
        want = (ctypes.c_int * 5)()
        ws = oext(qlen, q.tobytes(), tlen, t.tobytes(), 5, bytes(opts[140:165]), kw["o_del"], kw["e_del"], kw["o_ins"], kw["e_ins"], w, end_bonus, zdrop, h0,
                  ctypes.byref(want, 0), ctypes.byref(want, 4), ctypes.byref(want, 8), ctypes.byref(want, 12), ctypes.byref(want, 16))
        got = (ctypes.c_int * 6)()
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        for force_lds in (0, 1, 2, 3, 4): # (4: the packed form with its rows in LDS, the one long reads take) the 32-bit register forms (when the query fits), the general LDS form, the production entry (diagonal certificate first,
                                          # packed 16-bit form for 64..126-base queries), and the same without the certificate
            assert units.unit_extend(q.tobytes(), qlen, t.tobytes(), tlen, ob, w, end_bonus, zdrop, h0, got, force_lds) == 0
            assert list(got) == [ws, want[0], want[1], want[2], want[3], want[4]], (it, force_lds, qlen, tlen, kw, w, zdrop, h0)


def _check_extend_16bit_boundary(units, oracle, qlens=(126, 100, 65)):
    """the packed form of the two-chunk extension keeps its scores in 16-bit halves and must hand a query over to the 32-bit
    form when they could not fit (k_extend.hip: extend_pk2_ok, h0 + qlen * max score + 130 e_ins < 30000): match scores of 127
    and initial scores on either side of that line, with mismatches and a gap so that the rows are real DP"""
    oext = oracle.dll.o_ksw_extend2
    oext.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(5)
    base_opts = oracle.default_options()
    for qlen in qlens:
        for e_ins in (1, 3):
            edge = 30000 - qlen * 127 - 130 * e_ins
            for h0 in (edge - 1, edge, edge + 7, 200):
                kw = dict(a=127, b=100, o_del=60, e_del=10, o_ins=60, e_ins=e_ins)
                tlen = qlen + 40
                t = rng.integers(0, 4, size=tlen, dtype=np.uint8)
                q = t[:qlen].copy()
                for pos in rng.integers(5, qlen - 5, size=3):
                    q[pos] = (q[pos] + 1) % 4
                q = np.concatenate([q[:40], q[42:], rng.integers(0, 4, size=2, dtype=np.uint8)])      # a two-base deletion
                opts = B.set_opt(bytearray(base_opts), **kw)
                mat = []
                for i in range(4):
                    mat += [kw["a"] if i == j else -kw["b"] for j in range(4)] + [-1]
                mat += [-1] * 5
                B.set_opt(opts, mat=mat)
                want = (ctypes.c_int * 5)()
                ws = oext(qlen, q.tobytes(), tlen, t.tobytes(), 5, bytes(opts[140:165]), kw["o_del"], kw["e_del"], kw["o_ins"], kw["e_ins"], 100, 5, 0, h0,
                          ctypes.byref(want, 0), ctypes.byref(want, 4), ctypes.byref(want, 8), ctypes.byref(want, 12), ctypes.byref(want, 16))
                ob = ctypes.create_string_buffer(bytes(opts), 168)
                got = (ctypes.c_int * 6)()
                for mode in (0, 3, 4):
                    assert units.unit_extend(q.tobytes(), qlen, t.tobytes(), tlen, ob, 100, 5, 0, h0, got, mode) == 0
                    assert list(got) == [ws, want[0], want[1], want[2], want[3], want[4]], (qlen, e_ins, h0, mode, list(got), ws, list(want))
                assert ws > 5000                                       # (the scores really are beyond what a byte or a careless 16-bit sum holds)


def test_units_emu_sort(oracle):
    _check_sort(_load("emu"), oracle, (2, 3, 4, 9, 16, 17, 18, 33, 100, 600))


def test_units_emu_extend(oracle):
    _check_extend(_load("emu"), oracle, 40, 230)
    _check_extend_16bit_boundary(_load("emu"), oracle, qlens=(126, 65))


@pytest.mark.gpu
def test_units_gpu_sort(oracle):
    _check_sort(_load("hip"), oracle, (2, 3, 4, 9, 16, 17, 18, 33, 100, 600, 5000))


@pytest.mark.gpu
def test_units_gpu_extend(oracle):
    _check_extend(_load("hip"), oracle, 1500, 300)
    _check_extend_16bit_boundary(_load("hip"), oracle)


def _flt_reference(qb, qe, w, alt, mask_level, drop_ratio, max_chain_gap, min_seed_len):
    """mem_chain_flt's overlap loop (upstream bwamem.c, the loop over the weight-sorted chains), restated in Python with C's
    float comparisons -> (kept flag per chain, first shadowed chain per kept chain)"""
    f32 = np.float32
    n = len(qb)
    kept = [0] * n; kept[0] = 3
    chains = [0]; first = [-1]
    for i in range(1, n):
        large = 0
        for idx, j in enumerate(chains):
            b_max, e_min = max(qb[j], qb[i]), min(qe[j], qe[i])
            if e_min > b_max and (not alt[j] or alt[i]):
                min_l = min(qe[i] - qb[i], qe[j] - qb[j])
                if f32(e_min - b_max) >= f32(min_l) * f32(mask_level) and min_l < max_chain_gap:
                    large = 1
                    if first[idx] < 0:
                        first[idx] = i
                    if f32(w[i]) < f32(w[j]) * f32(drop_ratio) and w[j] - w[i] >= min_seed_len << 1:
                        break
        else:
            chains.append(i); first.append(-1)
            kept[i] = 2 if large else 3
    return kept, first


def _check_chain_flt(units, oracle, n_cases, max_n):
    units.unit_chain_flt.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 2
    rng = np.random.default_rng(23)
    for it in range(n_cases):
        n = int(rng.integers(1, max_n))
        L = int(rng.choice([150, 150, 250, 1000]))
        style = it % 3
        if style == 0:            # a repeat family: near-identical spans and weights, everything overlaps
            qb = rng.integers(0, 12, size=n); qe = L - rng.integers(0, 12, size=n)
            w = np.sort(rng.integers(L - 60, L - 20, size=n))[::-1]
        elif style == 1:          # scattered short chains: few overlaps, nearly all kept
            qb = rng.integers(0, L - 20, size=n); qe = np.minimum(qb + rng.integers(19, 60, size=n), L)
            w = np.sort(rng.integers(19, 60, size=n))[::-1]
        else:                     # a few dominant chains shadowing many weak ones
            qb = rng.integers(0, L // 2, size=n); qe = np.minimum(qb + rng.integers(19, L, size=n), L)
            w = np.sort(np.where(rng.random(n) < 0.1, rng.integers(100, 150, size=n), rng.integers(19, 70, size=n)))[::-1]
        alt = (rng.random(n) < rng.choice([0.0, 0.0, 0.3])).astype(np.int32)
        qb, qe, w = (np.ascontiguousarray(x, dtype=np.int32) for x in (qb, qe, w))
        kw = dict(mask_level=float(rng.choice([0.5, 0.5, 0.2, 0.9])), drop_ratio=float(rng.choice([0.5, 0.5, 0.8])), max_chain_gap=int(rng.choice([10000, 10000, 100])))
        opts = B.set_opt(bytearray(oracle.default_options()), **kw)
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        want_kept, want_first = _flt_reference(qb.tolist(), qe.tolist(), w.tolist(), alt.tolist(), kw["mask_level"], kw["drop_ratio"], kw["max_chain_gap"], 19)
        for wave in (0, 1):
            got_kept = np.full(n, -9, dtype=np.int32); got_first = np.full(n, -9, dtype=np.int32)
            nk = units.unit_chain_flt(ob, n, qb.ctypes.data, qe.ctypes.data, w.ctypes.data, alt.ctypes.data, wave, got_kept.ctypes.data, got_first.ctypes.data)
            assert nk == len(want_first), (it, wave, n, nk, len(want_first))
            assert got_kept.tolist() == want_kept and got_first[:nk].tolist() == want_first, (it, wave, n, kw)


def test_units_emu_chain_flt(oracle):
    _check_chain_flt(_load("emu"), oracle, 24, 200)


@pytest.mark.gpu
def test_units_gpu_chain_flt(oracle):
    _check_chain_flt(_load("hip"), oracle, 300, 1500)


def _check_sort_regs(units, n_cases, max_n):
    """the regions sorted through key records must come out in the very permutation the sort of the regions themselves gives
    (ties included: the record index travels in `pad_`)"""
    assert units.unit_sizeof_alnreg() == 96
    reg = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("mid", "<i4", 10),
                    ("n_comp", "<i4"), ("is_alt", "<i4"), ("frac_rep", "<f4"), ("pad_", "<i4"), ("hash", "<u8")])
    assert reg.itemsize == 96
    units.unit_sort_regs.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(41)
    for it in range(n_cases):
        n = int(rng.integers(1, max_n))
        a = np.zeros(n, dtype=reg)
        few = int(rng.choice([2, 5, 1000000]))                                  # many ties ... hardly any
        a["re"] = rng.integers(-3, few, size=n); a["rb"] = rng.integers(0, few, size=n); a["qb"] = rng.integers(-2, min(few, 150), size=n)
        a["score"] = rng.integers(-1, min(few, 151), size=n); a["is_alt"] = rng.integers(0, 2, size=n)
        a["hash"] = rng.integers(0, few, size=n, dtype=np.uint64) << np.uint64(int(rng.choice([0, 40, 63])))
        a["pad_"] = np.arange(n)
        for which in range(4):
            x, y = a.copy(), a.copy()
            assert units.unit_sort_regs(n, x.ctypes.data, which, 0) == 0 and units.unit_sort_regs(n, y.ctypes.data, which, 1) == 0
            assert x.tobytes() == y.tobytes(), (it, n, which, few)
            assert sorted(x["pad_"].tolist()) == list(range(n))


def test_units_emu_sort_regs():
    _check_sort_regs(_load("emu"), 12, 300)


@pytest.mark.gpu
def test_units_gpu_sort_regs():
    _check_sort_regs(_load("hip"), 150, 3000)


REG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("sub", "<i4"),
                      ("alt_sc", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"),
                      ("secondary_all", "<i4"), ("seedlen0", "<i4"), ("n_comp", "<i4"), ("is_alt", "<i4"), ("frac_rep", "<f4"), ("pad_", "<i4"), ("hash", "<u8")])


def _tandem_regions(rng, n, tie_level):
    """hit lists the way a mate inside a tandem array carries them: clusters of near-identical regions (same or neighbouring
    copy of the period, same or shifted stretch of the read), scores from a narrow range, many shared end coordinates"""
    a = np.zeros(n, dtype=REG_DTYPE)
    period = int(rng.choice([171, 50, 684]))
    n_clu = max(1, n // int(rng.choice([1, 2, 4])))
    c_rb = 100000 + period * rng.integers(0, max(2, n_clu // 2), size=n_clu) + rng.integers(0, 3, size=n_clu) * (tie_level < 2)
    c_len = rng.integers(40, 101, size=n_clu)
    c_qb = np.where(rng.random(n_clu) < 0.5, 0, rng.integers(0, 60, size=n_clu))
    k = rng.integers(0, n_clu, size=n)
    jit = (lambda: rng.integers(-2, 3, size=n) * (rng.random(n) < (0.1, 0.4, 0.8)[tie_level]))
    ln = np.maximum(20, c_len[k] + jit())
    a["rb"] = c_rb[k] + jit(); a["re"] = a["rb"] + ln + jit() * (rng.random(n) < 0.2)
    a["qb"] = np.maximum(0, c_qb[k] + jit()); a["qe"] = a["qb"] + ln
    a["rid"] = (rng.random(n) < 0.03).astype(np.int32)
    a["rb"] += 50000000 * a["rid"]; a["re"] += 50000000 * a["rid"]             # (a contig is a stretch of the coordinate axis)
    a["score"] = ln - 5 * rng.integers(0, (2, 3, 6)[tie_level], size=n)
    a["truesc"] = a["score"]; a["csub"] = rng.integers(0, 60, size=n); a["seedcov"] = ln // 2; a["secondary"] = -1
    a["n_comp"] = rng.integers(0, 4, size=n); a["w"] = rng.integers(0, 100, size=n)
    a["pad_"] = np.arange(n)                                                  # identity of the record: which of two twins survives shows
    return a


def _check_matesw_list(units, n_cases, max_n0, max_add):
    """the mate's hit list after a sequence of rescued regions: matesw_insert (post_common.h) must leave exactly the records
    upstream's insert + mem_sort_dedup_patch leaves, in the same order"""
    assert units.unit_sizeof_alnreg() == REG_DTYPE.itemsize == 96
    units.unit_matesw_list.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    orc = B.oracle_lib()
    rng = np.random.default_rng(77)
    fast = slow = 0
    for it in range(n_cases):
        n0 = int(rng.integers(0, max_n0)) if it else max_n0                    # the first case is the largest
        n_add = int(rng.integers(1, max_add))
        tie_level = it % 3
        both = _tandem_regions(rng, n0 + n_add, tie_level)
        sel = rng.permutation(n0 + n_add)
        a0, add = both[sel[:n0]].copy(), both[sel[n0:]].copy()
        opts = B.set_opt(bytearray(orc.default_options()), max_chain_gap=int(rng.choice([10000, 10000, 100])), mask_level_redun=float(rng.choice([0.95, 0.95, 0.5])))
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        res = []
        for incr in (0, 1, 2):                                                 # upstream's sequence, the lane's short cut, the wavefront's
            buf = np.zeros(n0 + n_add, dtype=REG_DTYPE); buf[:n0] = a0
            stat = (ctypes.c_int * 3)()
            assert units.unit_matesw_list(ob, buf.ctypes.data, n0, add.ctypes.data, n_add, incr, stat) == 0
            res.append((stat[0], buf[:stat[0]].tobytes(), stat[1], stat[2]))
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1], (it, n0, n_add, tie_level, res[0][0], res[1][0])
        assert res[0][0] == res[2][0] and res[0][1] == res[2][1], ("wave", it, n0, n_add, tie_level, res[0][0], res[2][0])
        assert res[0][3] == n_add and res[1][2:] == res[2][2:]
        fast += n_add - res[1][3]; slow += res[1][3]
    assert fast > 4 * slow, (fast, slow)                                       # the short cut is what normally runs
    return fast, slow


def test_units_emu_matesw_list():
    _check_matesw_list(_load("emu"), 60, 300, 40)


@pytest.mark.gpu
def test_units_gpu_matesw_list():
    _check_matesw_list(_load("hip"), 120, 2500, 150)


class _KswR(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


def _check_sw_jobs(units, oracle, n_jobs, seed):
    """ksw_align2 as mate rescue (byte mode: two alignments per lane in packed halves for up to 160-base mates, one per 16-lane group
    beyond) and seed re-scoring (16-bit mode) run it, job lists through launch_sw_jobs, against the oracle's o_ksw_align2: score,
    end points, second best, and the start points of the reverse pass"""
    oalign = oracle.dll.o_ksw_align2
    oalign.restype = _KswR
    oalign.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p] + [ctypes.c_int] * 5
    units.unit_sw_jobs.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4
    rng = np.random.default_rng(seed)
    opts = oracle.default_options()
    ob = ctypes.create_string_buffer(bytes(opts), 168)
    mat = bytes(opts[140:165])
    target = rng.integers(0, 4, size=60000, dtype=np.uint8)
    target[20000:20900] = np.tile(rng.integers(0, 4, size=45, dtype=np.uint8), 20)         # a tandem stretch: second-best hits, ties
    qs, qoff, toff, tlen, xtra = [], [0], [], [], []
    for it in range(n_jobs):
        l = int(rng.choice([150, 150, 150, 100, 76, 30, 200, 249, 250, 256]))              # (beyond 256 bases the lane-scalar form runs: not a job kernel)
        tl = int(rng.integers(max(30, l // 2), 900))
        t0 = int(rng.integers(0, len(target) - tl))
        if it % 5 == 0:
            t0 = int(rng.integers(19900, 20500))
        kind = it % 4
        if kind == 3:                                                                   # unrelated query
            q = rng.integers(0, 4, size=l, dtype=np.uint8)
        else:                                                                           # a mutated stretch of the window (sometimes hanging over its end)
            s = int(rng.integers(0, max(1, tl - l // 2)))
            q = np.resize(target[t0 + s: t0 + s + l], l).copy()
            mut = rng.random(l) < rng.choice([0.0, 0.02, 0.1])
            q[mut] = rng.integers(0, 5, size=int(mut.sum()), dtype=np.uint8)
            if rng.random() < 0.3 and l > 20:
                cut = int(rng.integers(5, l - 5)); k = int(rng.integers(1, 6))
                q = np.concatenate([q[:cut], q[cut + k:], rng.integers(0, 4, size=k, dtype=np.uint8)])
        x = 0x40000 | 0x80000 | (0x10000 if l * 1 < 250 else 0) | 19                    # KSW_XSUBO | KSW_XSTART | (KSW_XBYTE) | min_seed_len * a
        if it % 7 == 6:
            x &= ~0x80000                                                               # seed re-scoring asks for the first pass only
        qs.append(q); qoff.append(qoff[-1] + l); toff.append(t0); tlen.append(tl); xtra.append(x)
    qcat = np.concatenate(qs).astype(np.uint8)
    qoff, toff, tlen, xtra = (np.asarray(v, dtype=t) for v, t in ((qoff, np.int64), (toff, np.int64), (tlen, np.int32), (xtra, np.int32)))
    out = np.zeros((n_jobs, 7), dtype=np.int32)
    rc = units.unit_sw_jobs(ob, n_jobs, qcat.ctypes.data, qoff.ctypes.data, target.ctypes.data, len(target), toff.ctypes.data, tlen.ctypes.data, xtra.ctypes.data, out.ctypes.data)
    assert rc == 0, rc
    for i in range(n_jobs):
        w = oalign(len(qs[i]), qs[i].tobytes(), int(tlen[i]), target[toff[i]: toff[i] + tlen[i]].tobytes(), 5, mat, 6, 1, 6, 1, int(xtra[i]))
        want = [w.score, w.te, w.qe, w.score2, w.te2, w.tb, w.qb]
        assert out[i].tolist() == want, (i, len(qs[i]), int(tlen[i]), hex(int(xtra[i])), out[i].tolist(), want)


def test_units_emu_sw_jobs(oracle):
    _check_sw_jobs(_load("emu"), oracle, 41, 7)


@pytest.mark.gpu
def test_units_gpu_sw_jobs(oracle):
    _check_sw_jobs(_load("hip"), oracle, 3001, 8)


def _check_extend_long(units, oracle, n_cases, max_q, seed=19):
    """queries beyond the register forms (the band slides along rows kept in LDS rings): the general form and the packed form with
    its lane-private pair rings, on long extensions with substitutions and indels, bands of 100 and 200"""
    oext = oracle.dll.o_ksw_extend2
    oext.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(seed)
    opts = oracle.default_options()
    ob = ctypes.create_string_buffer(bytes(opts), 168)
    for it in range(n_cases):
        qlen = int(rng.integers(200, max_q)); tlen = qlen + int(rng.integers(-60, 200))
        t = rng.integers(0, 4, size=max(tlen, 10), dtype=np.uint8); tlen = len(t)
        rate = float(rng.choice([0.01, 0.05, 0.14]))
        q = []
        i = 0
        while len(q) < qlen and i < tlen:                           # ONT-like: substitutions, single-base insertions and deletions
            r = rng.random()
            if r < rate / 4: i += 1; continue
            if r < rate / 2: q.append(int(rng.integers(0, 4))); continue
            q.append(int((t[i] + (1 + rng.integers(0, 3)) * (rng.random() < rate)) % 4)); i += 1
        q = np.array(q + [int(x) for x in rng.integers(0, 4, size=qlen - len(q))], dtype=np.uint8)
        if it % 4 == 3:
            q[len(q) // 2:] = rng.integers(0, 4, size=len(q) - len(q) // 2, dtype=np.uint8)     # the second half unrelated: z-drop / early end
        w = int(rng.choice([100, 200])); zdrop = int(rng.choice([100, 100, 0])); h0 = int(rng.integers(19, 200))
        want = (ctypes.c_int * 5)()
        ws = oext(qlen, q.tobytes(), tlen, t.tobytes(), 5, bytes(opts[140:165]), 6, 1, 6, 1, w, 5, zdrop, h0,
                  ctypes.byref(want, 0), ctypes.byref(want, 4), ctypes.byref(want, 8), ctypes.byref(want, 12), ctypes.byref(want, 16))
        got = (ctypes.c_int * 6)()
        for mode in (1, 2, 4):
            assert units.unit_extend(q.tobytes(), qlen, t.tobytes(), tlen, ob, w, 5, zdrop, h0, got, mode) == 0
            assert list(got) == [ws, want[0], want[1], want[2], want[3], want[4]], (it, mode, qlen, tlen, w, zdrop, h0, list(got), ws, list(want))


def test_units_emu_extend_long(oracle):
    _check_extend_long(_load("emu"), oracle, 6, 900)


@pytest.mark.gpu
def test_units_gpu_extend_long(oracle):
    _check_extend_long(_load("hip"), oracle, 120, 6000)


def _check_global(units, oracle, n_jobs, max_len, seed=23):
    """ksw_global2 with traceback as k_gcigar's wave forms run it (band across the lanes, one diagonal per slot): the 32-bit form and
    the packed 16-bit form (two chunks per instruction stream, values re-based every 64 rows) against the oracle's o_ksw_global2 --
    score and CIGAR; related and unrelated sequences (scores far below zero: the re-basing), N bases, bands of 33 to 401 columns
    either side, three option sets (bwa's default, -x ont2d's, one the packed form must refuse), and the give-up path of the range check"""
    og = oracle.dll.o_ksw_global2
    og.restype = ctypes.c_int
    og.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    units.unit_global.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    rng = np.random.default_rng(seed)
    target = rng.integers(0, 4, size=max_len * 4 + 4000, dtype=np.uint8)
    for oi, (a, b, o, e) in enumerate([(1, 4, 6, 1), (1, 1, 1, 1), (2, 20, 30, 3)]):
        scmat = [(a if i == j else -b) if i < 4 and j < 4 else -1 for i in range(5) for j in range(5)]     # bwa_fill_scmat
        opts = B.set_opt(oracle.default_options(), a=a, b=b, o_del=o, e_del=e, o_ins=o, e_ins=e, mat=scmat)
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        mat = bytes(opts[140:165])
        qs, qoff, toff, tlen, wv = [], [0], [], [], []
        for it in range(n_jobs):
            w = int(rng.choice([16, 33, 50, 100, 100, 100, 150, 200, 400]))
            tl = int(rng.integers(40, max_len)) if it % 3 else int(rng.integers(max_len // 2, max_len))
            t0 = int(rng.integers(0, len(target) - tl))
            t = target[t0: t0 + tl]
            if it % 6 == 5:                                                             # unrelated: the score sinks by ~3 a row
                q = rng.integers(0, 4, size=max(8, tl + int(rng.integers(-w // 3, w // 3 + 1))), dtype=np.uint8)
            else:                                                                       # a noisy copy: substitutions, N, single-base and longer indels
                sub, ind = rng.choice([0.0, 0.02, 0.08, 0.15]), rng.choice([0.0, 0.005, 0.03])
                out = []
                j = 0
                while j < tl:
                    r = rng.random()
                    if r < ind: out.append(int(rng.integers(0, 4)))                    # insertion
                    elif r < 2 * ind: j += 1 + (int(rng.integers(0, 12)) if rng.random() < 0.1 else 0)   # deletion
                    else:
                        c = int(t[j]); j += 1
                        if rng.random() < sub: c = int(rng.integers(0, 5))
                        out.append(c)
                q = np.asarray(out[: tl + w // 2] if len(out) > 8 else [0] * 9, dtype=np.uint8)
                if len(q) < tl - w // 2: q = np.concatenate([q, rng.integers(0, 4, size=tl - w // 2 - len(q), dtype=np.uint8)])
            d = abs(len(q) - tl)
            w = max(w, d + 3)                                                           # bwa_gen_cigar2 never asks for less
            if (2 * w + 1 + 63) // 64 > 13: w = 400
            if abs(len(q) - tl) > w: q = np.resize(q, tl)
            qs.append(q); qoff.append(qoff[-1] + len(q)); toff.append(t0); tlen.append(tl); wv.append(w)
        qcat = np.concatenate(qs).astype(np.uint8)
        qoffa, toffa, tlena, wva = (np.asarray(v, dtype=t) for v, t in ((qoff, np.int64), (toff, np.int64), (tlen, np.int32), (wv, np.int32)))
        cap = 2 * max_len + 16
        want = []
        for i in range(n_jobs):
            nc = ctypes.c_int(0); cg = ctypes.POINTER(ctypes.c_uint32)()
            sc = og(len(qs[i]), qs[i].tobytes(), int(tlen[i]), target[toff[i]: toff[i] + tlen[i]].tobytes(), 5, mat, o, e, o, e, int(wv[i]), ctypes.byref(nc), ctypes.byref(cg))
            want.append((sc, [cg[k] for k in range(nc.value)]))
            libc.free(cg)
        for mode, rng_over in ((0, 0), (1, 0), (1, 40)):
            out = np.zeros((n_jobs, 4), dtype=np.int32)
            cig = np.zeros((n_jobs, cap), dtype=np.uint32)
            rc = units.unit_global(ob, n_jobs, qcat.ctypes.data, qoffa.ctypes.data, target.ctypes.data, len(target), toffa.ctypes.data, tlena.ctypes.data, wva.ctypes.data,
                                   mode, rng_over, cap, out.ctypes.data, cig.ctypes.data)
            assert rc == 0, rc
            for i in range(n_jobs):
                got = (int(out[i, 0]), cig[i, : out[i, 1]].tolist())
                assert out[i, 3] == 0 and got == want[i], (oi, mode, rng_over, i, len(qs[i]), int(tlen[i]), int(wv[i]), out[i].tolist(), got[0], want[i][0], got[1][:8], want[i][1][:8])
            nch = (2 * wva + 1 + 63) // 64
            if mode == 1 and oi < 2 and rng_over == 0:
                assert (out[nch >= 2, 2] == 1).all(), (oi, out[:, 2].tolist())          # the packed form took every job it is built for
            if mode == 1 and oi == 2:
                assert (out[:, 2] == 0).all()                                            # penalties this large: refused up front
            if mode == 1 and rng_over:
                assert oi == 2 or (out[nch >= 2, 2] == 2).any()                          # a 40-point range: most jobs give up at a check and run the 32-bit form


def test_units_emu_global(oracle):
    _check_global(_load("emu"), oracle, 6, 300)


@pytest.mark.gpu
def test_units_gpu_global(oracle):
    _check_global(_load("hip"), oracle, 240, 2600)
    _check_global(_load("hip"), oracle, 12, 12000, seed=5)
