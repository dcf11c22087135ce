"""The reference's CPU-runnable case: benchmarks/bitpackingbenchmark.c packs and
unpacks ONE 128-integer block per width b = 1..32, 500 times, with data from
libc rand() masked to b bits (:34-42) and, for the d1 demo, running sums of
such values (:44-53); calls simdpackwithoutmask / simdunpack (:73,:84) and
simdpackwithoutmaskd1(0, ...) / simdunpackd1(0, ...) (:121,:132).  The program
itself checks nothing, so the same data and the same call sequence are checked
here: restatement against the compiled reference on the CPU, and the library
(host signatures and the batched device form) against the restatement on the
GPU."""
import ctypes as C

import numpy as np
import pytest

LENGTH = 128
REPEAT = 500


def benchmark_blocks():
    """[(bit, plain block, d1 block)] drawn as the benchmark draws them (srand default seed)."""
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    libc.srand(1)
    out = []
    for bit in range(1, 33):
        mask = (1 << bit) - 1
        plain = np.array([libc.rand() & mask for _ in range(LENGTH)], dtype=np.uint32)
        out.append([bit, plain, None])
    for rec in out:
        mask = (1 << rec[0]) - 1
        gaps = np.array([libc.rand() & mask for _ in range(LENGTH)], dtype=np.uint64)
        rec[2] = (np.cumsum(gaps) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return out


def test_restatement_matches_reference_on_benchmark_blocks(orc, ref):
    for bit, plain, sorted_ in benchmark_blocks():
        a, b = orc.simdpackwithoutmask(plain, bit), ref.simdpackwithoutmask(plain, bit)
        assert np.array_equal(a, b), bit
        assert np.array_equal(orc.simdunpack(a, bit), ref.simdunpack(b, bit))
        assert np.array_equal(ref.simdunpack(b, bit), plain)
        a, b = orc.simdpackwithoutmaskd1(0, sorted_, bit), ref.simdpackwithoutmaskd1(0, sorted_, bit)
        assert np.array_equal(a, b), bit
        assert np.array_equal(orc.simdunpackd1(0, a, bit), ref.simdunpackd1(0, b, bit))
        assert np.array_equal(ref.simdunpackd1(0, b, bit), sorted_)  # sums that wrap mod 2^32 decode the same way


@pytest.mark.gpu
def test_library_host_calls_on_benchmark_blocks(orc):
    from simdcomp_amd import binding

    lib = binding.lib()
    for bit, plain, sorted_ in benchmark_blocks():
        buf = np.zeros(LENGTH * 4, np.uint8)
        back = np.zeros(LENGTH, np.uint32)
        for _ in range(3):  # the benchmark repeats each call; repeating must not change the result
            lib.simdpackwithoutmask(plain.ctypes.data, buf.ctypes.data, bit)
        assert np.array_equal(buf[:16 * bit], orc.simdpackwithoutmask(plain, bit)[:16 * bit]), bit
        for _ in range(3):
            lib.simdunpack(buf.ctypes.data, back.ctypes.data, bit)
        assert np.array_equal(back, plain), bit
        buf[:] = 0
        for _ in range(3):
            lib.simdpackwithoutmaskd1(0, sorted_.ctypes.data, buf.ctypes.data, bit)
        assert np.array_equal(buf[:16 * bit], orc.simdpackwithoutmaskd1(0, sorted_, bit)[:16 * bit]), bit
        for _ in range(3):
            lib.simdunpackd1(0, buf.ctypes.data, back.ctypes.data, bit)
        assert np.array_equal(back, orc.simdunpackd1(0, buf, bit)), bit
        assert np.array_equal(back, sorted_), bit


@pytest.mark.gpu
def test_library_batched_calls_on_benchmark_blocks(orc):
    """The benchmark's 500 repetitions of one block as ONE batched call of 500 blocks per width."""
    import torch

    from simdcomp_amd import codec

    dev = torch.device("cuda:0")
    for bit, plain, sorted_ in benchmark_blocks():
        x = torch.from_numpy(np.tile(plain, REPEAT).view(np.int32)).to(dev)
        pay = codec.pack(x, bit, masked=False)
        want = orc.simdpackwithoutmask(plain, bit)[:16 * bit]
        assert np.array_equal(pay.cpu().numpy().reshape(REPEAT, 16 * bit), np.tile(want, (REPEAT, 1))), bit
        assert torch.equal(codec.unpack(pay, REPEAT, bit), x), bit
        y = torch.from_numpy(np.tile(sorted_, REPEAT).view(np.int32)).to(dev)
        seeds = torch.zeros(REPEAT, dtype=torch.int32, device=dev)
        pay = codec.packd1(y, bit, initvalues=seeds, masked=False)
        want = orc.simdpackwithoutmaskd1(0, sorted_, bit)[:16 * bit]
        assert np.array_equal(pay.cpu().numpy().reshape(REPEAT, 16 * bit), np.tile(want, (REPEAT, 1))), bit
        assert torch.equal(codec.unpackd1(seeds, pay, REPEAT, bit), y), bit
