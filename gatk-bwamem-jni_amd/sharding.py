"""Read-level sharding of one logical createAlignments call across ranks.

Reads are independent units, so a call's reads are cut into contiguous ranges on pair boundaries,
one range per GPU/rank, with NO data-path collective; responses are concatenated in input order.
For bit-parity every shard carries the index of its first read within the logical call
(read_id0), because the primary-marking tie-break hashes that index (upstream hash_64(id+i),
reached from jnibwa.c:214 with n_processed = 0).

The one exchange the path has: a paired-end call with inferred insert-size statistics (pes == NULL).
mem_pestat reduces over ALL pairs of the call between region finding and pairing, so each shard runs
phase 1, the per-pair (orientation, insert size) candidates are all-gathered (9 bytes per pair, over
whatever backend the process group uses: RCCL on GPUs, gloo in the CPU tests), every rank reduces the same
list to the same statistics, and phase 2 runs with them.  With statistics supplied by the caller there is
no exchange.
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


def _gather_candidates(dist, dir_b, is_b):
    """all-gather of the shards' candidate arrays (ragged: sizes first, then padded tensors)"""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    world = dist.get_world_size()
    n = torch.tensor([len(dir_b)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    d = torch.zeros(cap, dtype=torch.int8, device=dev); i = torch.zeros(cap, dtype=torch.int64, device=dev)
    if len(dir_b):
        d[:len(dir_b)] = torch.frombuffer(bytearray(dir_b), dtype=torch.int8).to(dev)
        i[:len(dir_b)] = torch.frombuffer(bytearray(is_b), dtype=torch.int64).to(dev)
    ds = [torch.zeros(cap, dtype=torch.int8, device=dev) for _ in range(world)]
    iss = [torch.zeros(cap, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ds, d); dist.all_gather(iss, i)
    all_d = torch.cat([ds[r][:sizes[r]] for r in range(world)]).cpu().numpy().tobytes()
    all_i = torch.cat([iss[r][:sizes[r]] for r in range(world)]).cpu().numpy().tobytes()
    return all_d, all_i, sum(sizes)


def align_shard(dll, idx, opts, reads, rank, world, paired=False, pes=None, dist=None):
    """align this rank's slice of `reads` through the device-level C ABI; returns the raw response bytes of the slice.
    paired with pes=None (statistics inferred): `dist` (an initialised torch.distributed) carries the one exchange."""
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
        if paired:
            struct.pack_into("<i", ob, 60, struct.unpack_from("<i", ob, 60)[0] | 0x2)          # MEM_F_PE
        if paired and pes is None and world > 1:
            dll.bwamem_hip_batch_pe_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
            dll.bwamem_hip_batch_pe_candidates.restype = ctypes.c_size_t
            dll.bwamem_hip_batch_pe_candidates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            dll.bwamem_hip_pestat.restype = None
            dll.bwamem_hip_pestat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            dll.bwamem_hip_batch_pe_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            if dll.bwamem_hip_batch_pe_begin(idx, ob, batch, b) != 0:
                raise RuntimeError("paired-end phase 1 failed")
            n = dll.bwamem_hip_batch_pe_candidates(batch, None, None)
            dbuf = ctypes.create_string_buffer(max(n, 1)); ibuf = ctypes.create_string_buffer(max(8 * n, 8))
            dll.bwamem_hip_batch_pe_candidates(batch, dbuf, ibuf)
            all_d, all_i, n_all = _gather_candidates(dist, dbuf.raw[:n], ibuf.raw[:8 * n])
            pbuf = ctypes.create_string_buffer(128)                                               # mem_pestat_t[4]
            dll.bwamem_hip_pestat(ob, all_d, all_i, n_all, pbuf)
            if dll.bwamem_hip_batch_pe_finish(idx, ob, pbuf, batch) != 0:
                raise RuntimeError("paired-end phase 2 failed")
        elif dll.bwamem_hip_batch_align(idx, ob, ctypes.create_string_buffer(pes, len(pes)) if pes is not None else None, batch, b) != 0:
            raise RuntimeError("align failed")
        n = dll.bwamem_hip_batch_result_bytes(batch)
        out = ctypes.create_string_buffer(max(n, 1))
        if dll.bwamem_hip_batch_download(batch, out) != 0:
            raise RuntimeError("download failed")
        return out.raw[:n]
    finally:
        dll.bwamem_hip_batch_free(batch)
