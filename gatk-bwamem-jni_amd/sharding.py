"""Read-level sharding of one logical createAlignments call across ranks.

Reads are independent units, so a call's reads are cut into contiguous ranges on pair boundaries,
one range per GPU/rank, with NO data-path collective; responses are concatenated in input order.
For bit-parity every shard carries the index of its first read within the logical call
(read_id0), because the primary-marking tie-break hashes that index (upstream hash_64(id+i),
reached from jnibwa.c:214 with n_processed = 0).
"""
import ctypes
import struct


def shard_range(n_reads, rank, world, paired=False):
    """contiguous [begin, end) of reads for this rank; pairs are never split"""
    unit = 2 if paired else 1
    n_units = n_reads // unit
    base, extra = divmod(n_units, world)
    b = rank * base + min(rank, extra)
    e = b + base + (1 if rank < extra else 0)
    if rank == world - 1 and not paired:
        return b * unit, n_reads
    return b * unit, e * unit


def align_shard(dll, idx, opts, reads, rank, world, paired=False):
    """align this rank's slice of `reads` through the device-level C ABI; returns the raw response bytes of the slice"""
    b, e = shard_range(len(reads), rank, world, paired)
    mine = reads[b:e]
    req = struct.pack("<i", len(mine)) + b"".join(r + b"\0" for r in mine)
    dll.bwamem_hip_batch_upload.restype = ctypes.c_void_p
    dll.bwamem_hip_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    dll.bwamem_hip_batch_align.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    dll.bwamem_hip_batch_result_bytes.restype = ctypes.c_size_t
    dll.bwamem_hip_batch_result_bytes.argtypes = [ctypes.c_void_p]
    dll.bwamem_hip_batch_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    dll.bwamem_hip_batch_free.argtypes = [ctypes.c_void_p]
    batch = dll.bwamem_hip_batch_upload(idx, req, len(req))
    if not batch:
        raise RuntimeError("batch upload failed")
    try:
        ob = ctypes.create_string_buffer(bytes(opts), 168)
        if dll.bwamem_hip_batch_align(idx, ob, None, batch, b) != 0:
            raise RuntimeError("align failed")
        n = dll.bwamem_hip_batch_result_bytes(batch)
        out = ctypes.create_string_buffer(max(n, 1))
        if dll.bwamem_hip_batch_download(batch, out) != 0:
            raise RuntimeError("download failed")
        return out.raw[:n]
    finally:
        dll.bwamem_hip_batch_free(batch)
