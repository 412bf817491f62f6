"""N>1 path on CPU: two gloo ranks each align their contiguous shard (with its global read_id0)
and the concatenation must equal the single-call response.  The device back end here is the
emulation build (test infrastructure); on a GPU box the same code path runs on libbwamem_hip.so."""
import os
import sys

import pytest
import torch.multiprocessing as mp

import bwalib as B

sys.path.insert(0, B.PKG)


def _worker(rank, world, port, img, reads, q, paired=False):
    import torch.distributed as dist
    import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lib = B.product_lib(emu=True)
    h = lib.open_index(img)
    part = sharding.align_shard(lib.dll, h, lib.default_options(), reads, rank, world, paired=paired, dist=dist)
    parts = [None] * world
    dist.all_gather_object(parts, part)          # test-side gather only; the data path has no collective
    if rank == 0:
        q.put(b"".join(parts))
    dist.barrier()
    lib.destroy_index(h)
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    import sharding
    for n in (0, 1, 7, 10, 11):
        for w in (1, 2, 3, 8):
            cuts = [sharding.shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
    for r in range(3):
        b, e = sharding.shard_range(10, r, 3, paired=True)
        assert b % 2 == 0 and e % 2 == 0


def test_two_rank_shards_equal_single_call(oracle, small_genome):
    B.build_emu()
    seqs, img = small_genome
    # equal-score repeats make the hash tie-break (read index dependent) matter
    reads = B.simulate_reads(seqs, 14, length=100, seed=77, sub=0.01) + [seqs[0][1][5000:5100]] * 4
    ho = oracle.open_index(img)
    want = oracle.align_raw(ho, oracle.default_options(), B.pack_request(reads))
    oracle.destroy_index(ho)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, img, reads, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want


def test_two_rank_paired_shards_share_insert_size_statistics(oracle, small_genome):
    """paired-end with inferred statistics: mem_pestat reduces over all pairs of the call, so the shards exchange their
    candidates between the phases (the path's one collective); the concatenation must equal the single-call response.
    The insert sizes of the two halves differ on purpose: per-shard statistics would give different records."""
    B.build_emu()
    seqs, img = small_genome
    # first half: 24 pairs around 260; second half: 22 around 420 and 2 around 250 -- outside the bounds the second half would
    # infer on its own (about 290..580), inside the bounds of the whole call (quartiles 260 / 420)
    pairs = (B.simulate_pairs(seqs, 24, length=100, seed=5, ins_mean=260, ins_sd=12) + B.simulate_pairs(seqs, 22, length=100, seed=6, ins_mean=420, ins_sd=25)
             + B.simulate_pairs(seqs, 2, length=100, seed=8, ins_mean=250, ins_sd=3))
    opts = B.set_opt(oracle.default_options(), flag=B.get_opt(oracle.default_options(), "flag") | B.MEM_F_PE)
    ho = oracle.open_index(img)
    want = oracle.align_raw(ho, opts, B.pack_request(pairs))
    # the case must be able to tell: each half aligned as a call of its own (its own statistics) gives other records
    import ctypes
    fn = oracle.dll.oracle_createAlignmentsAt
    fn.restype = ctypes.c_void_p
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int64]
    halves = b""
    for lo in (0, len(pairs) // 2):
        req = B.pack_request(pairs[lo:lo + len(pairs) // 2])
        rb = ctypes.create_string_buffer(req, len(req)); sz = ctypes.c_size_t()
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        p = fn(ho, ob, None, rb, ctypes.byref(sz), lo)
        halves += ctypes.string_at(p, sz.value)
    assert halves != want
    oracle.destroy_index(ho)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, img, pairs, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want
