"""One BwaMemIndex handle over several devices (jnibwa_openIndex + BWAMEM_HIP_DEVICES, pipeline.cpp): a large
jnibwa_createAlignments call is cut across the replicas -- contiguous ranges on pair boundaries, each numbered from its
first read's index in the call, the insert-size candidates of all shards reduced together -- and small concurrent calls go to
the replicas in turn.  What GATK sees must not depend on any of it (jnibwa.c:197-235: one call = the whole batch;
BwaMemIndex.java:16-27: one index shared by all threads).  BWAMEM_HIP_DEVICES=0,0 puts two replicas on one device, which is how
a one-GPU box (and the emulation build here) exercises the path.  Own process: the knobs are read when the library first
sees a multi-replica handle."""
import os
import subprocess
import sys

import pytest

import bwalib as B

CODE = r'''
import os, sys, threading
sys.path.insert(0, %(tests)r)
import ctypes
import bwalib as B
lib, orc = B.product_lib(emu=%(emu)r), B.oracle_lib()
img = %(img)r
seqs = []
for blk in open(img[:-4]).read().split(">")[1:]:
    name, _, body = blk.partition("\n")
    seqs.append((name.strip(), body.replace("\n", "").encode()))
h, ho = lib.open_index(img), orc.open_index(img)
assert h, "openIndex failed"
lib.dll.bwamem_hip_index_replicas.argtypes = [ctypes.c_void_p]
assert lib.dll.bwamem_hip_index_replicas(h) == %(replicas)d
n = %(n)d
# single-end: equal-score repeats make the hash tie-break (read index within the call) matter
reads = B.simulate_reads(seqs, n, length=100, seed=77, sub=0.01) + [seqs[0][1][5000:5100]] * 5
opts = lib.default_options()
for rd in (reads, reads[:-1], reads[:3], reads[:1], []):
    req = B.pack_request(rd)
    assert lib.align_raw(h, opts, req) == orc.align_raw(ho, opts, req), ("single-end", len(rd))
# paired-end, statistics inferred: the halves of the call have different insert sizes, so per-shard statistics would show
pairs = (B.simulate_pairs(seqs, n // 2, length=100, seed=5, ins_mean=260, ins_sd=12) + B.simulate_pairs(seqs, n // 2 - 2, length=100, seed=6, ins_mean=420, ins_sd=25)
         + B.simulate_pairs(seqs, 2, length=100, seed=8, ins_mean=250, ins_sd=3))
po = B.set_opt(lib.default_options(), flag=B.MEM_F_PE)
want = orc.align_raw(ho, po, B.pack_request(pairs))
halves = b""
fn = orc.dll.oracle_createAlignmentsAt
fn.restype = ctypes.c_void_p
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int64]
for lo in (0, len(pairs) // 2):
    rq = B.pack_request(pairs[lo:lo + len(pairs) // 2])
    rb = ctypes.create_string_buffer(rq, len(rq)); sz = ctypes.c_size_t()
    ob = ctypes.create_string_buffer(bytes(po), 168)
    halves += ctypes.string_at(fn(ho, ob, None, rb, ctypes.byref(sz), lo), sz.value)
assert halves != want or n < 40                           # (the case can tell shared statistics from per-shard ones)
assert lib.align_raw(h, po, B.pack_request(pairs)) == want, "paired-end, inferred statistics"
odd = pairs[:-1]                                          # an odd read count: the last read has no mate and gives no bytes
assert lib.align_raw(h, po, B.pack_request(odd)) == orc.align_raw(ho, po, B.pack_request(odd)), "paired-end, odd count"
pes = B.pack_pestat(150, 450, 300.0, 30.0)
assert lib.align_raw(h, po, B.pack_request(pairs), pes) == orc.align_raw(ho, po, B.pack_request(pairs), pes), "paired-end, supplied statistics"
# small concurrent calls on the shared handle: each is one call of its own (read indices from 0), whichever replica takes it
os.environ["BWAMEM_TEST_NOTE"] = "concurrent"
small = [reads[i::4][:6] for i in range(4)]
wants = [orc.align_raw(ho, opts, B.pack_request(s)) for s in small]
got = [None] * 4
def call(i):
    got[i] = lib.align_raw(h, opts, B.pack_request(small[i]))
th = [threading.Thread(target=call, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
assert got == wants, "concurrent small calls"
lib.destroy_index(h); orc.destroy_index(ho)
print("multi-device-ok")
'''


def _run(emu, img, n, devices="0,0", replicas=2, split_min="1"):
    env = dict(os.environ, BWAMEM_HIP_DEVICES=devices, BWAMEM_HIP_SPLIT_MIN=split_min)
    code = CODE % dict(tests=os.path.join(B.ROOT, "tests"), emu=emu, img=img, n=n, replicas=replicas)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0 and "multi-device-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_two_replicas_on_the_emulation(small_genome):
    B.build_emu()
    seqs, img = small_genome
    _run(True, img, 48)


def test_small_calls_are_not_cut(small_genome):
    """below BWAMEM_HIP_SPLIT_MIN reads per replica a call goes to one replica whole (the default spares real callers'
    small batches the walk and the second device's launch overheads); three replicas, calls of a few reads"""
    B.build_emu()
    seqs, img = small_genome
    _run(True, img, 8, devices="0,0,0", replicas=3, split_min="1000000")


@pytest.mark.gpu
def test_two_replicas_on_one_gpu(small_genome):
    seqs, img = small_genome
    _run(False, img, 3000)


@pytest.mark.gpu
def test_three_replicas_medium_genome_on_one_gpu(medium_genome):
    seqs, img = medium_genome
    _run(False, img, 40000, devices="0,0,0", replicas=3)
