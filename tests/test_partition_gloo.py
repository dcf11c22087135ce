"""The N > 1 path on CPU: two gloo ranks take block ranges from the library's
partitioner, decode their shard (with the CPU oracle standing in for the
kernel), and the verification all-reduce of (checksum, mismatches) must equal
the single-process result.  Mirrors what bench.py does with RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simdcomp_amd import synth

M64 = (1 << 64) - 1


def checksum_np(words, index0=0):
    """Host statement of the position-sensitive checksum the library's kernel computes."""
    idx = (np.arange(len(words), dtype=np.uint64) + np.uint64(index0)) << np.uint64(32)
    return int(synth.splitmix64(idx ^ words.astype(np.uint64)).sum(dtype=np.uint64)) & M64


def _worker(rank, world, port, nblk, bit, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C

    import oracle
    from simdcomp_amd import binding

    lib = binding.lib()
    first, count = C.c_size_t(), C.c_size_t()
    assert lib.simdcomp_hip_partition(nblk, world, rank, C.byref(first), C.byref(count)) == 0
    f, c = first.value, count.value
    x = synth.sorted_gaps(3, nblk * 128, bit, first=1)
    offs = synth.block_offsets(x, 1)
    orc = oracle.Oracle()
    shard = x[f * 128:(f + c) * 128].copy()
    pay = orc.batch_pack("d1", shard, bit, seeds=offs[f:f + c].copy())
    dec = orc.batch_unpack("d1", pay, c, bit, seeds=offs[f:f + c].copy())
    acc = torch.tensor([checksum_np(dec, f * 128) >> 32, checksum_np(dec, f * 128) & 0xFFFFFFFF,
                        int((dec != shard).sum()), c], dtype=torch.int64)
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put(acc.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_range_shards():
    nblk, bit, world = 1001, 9, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nblk, bit, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = synth.sorted_gaps(3, nblk * 128, bit, first=1)
    # the two 32-bit halves were summed separately; recombine modulo 2^64
    whole = checksum_np(x)
    parts = [checksum_np(x[:((nblk * 1) // 2) * 128], 0), checksum_np(x[((nblk * 1) // 2) * 128:], ((nblk * 1) // 2) * 128)]
    assert (parts[0] + parts[1]) & M64 == whole
    assert ((got[0] << 32) + got[1]) & M64 == whole
    assert got[2] == 0 and got[3] == nblk
