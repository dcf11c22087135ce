"""Parity of the HIP kernels, called through the batched C ABI
(include/simdcomp_hip.h), against the CPU oracle and the golden fixtures.
Bit-exact: every comparison is array_equal on integers / bytes."""
import numpy as np
import pytest

import golden_util as G
from simdcomp_amd import synth

pytestmark = pytest.mark.gpu

BITS = list(range(33))
# ragged on purpose: not a multiple of the 32 blocks a workgroup handles per step
NBLK = 2 * 32 * 7 + 13


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from simdcomp_amd import codec

    return codec


def to_dev(a):
    import torch

    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).cuda()
    return torch.from_numpy(a).cuda()


def to_np(t, dtype=None):
    a = t.cpu().numpy()
    if a.dtype == np.int32:
        a = a.view(np.uint32)
    if a.dtype == np.int64:
        a = a.view(np.uint64)
    return a


@pytest.mark.parametrize("bit", BITS)
def test_plain_uniform(dev, orc, bit):
    for masked, x in ((False, synth.random_bits(1, NBLK * 128, bit)), (True, synth.rand_u32(2, NBLK * 128)),
                      (False, synth.rand_u32(3, NBLK * 128))):
        want = orc.batch_pack("plain", x, bit, masked=masked)
        got = to_np(dev.pack(to_dev(x), bit, masked=masked))
        assert np.array_equal(got, want), ("pack", bit, masked)
        dec = to_np(dev.unpack(to_dev(want), NBLK, bit))
        assert np.array_equal(dec, orc.batch_unpack("plain", want, NBLK, bit))
    # decoding arbitrary bytes
    junk = synth.rand_u32(4, NBLK * 4 * bit).view(np.uint8) if bit else np.zeros(0, np.uint8)
    assert np.array_equal(to_np(dev.unpack(to_dev(junk), NBLK, bit)), orc.batch_unpack("plain", junk, NBLK, bit))


@pytest.mark.parametrize("bit", BITS)
def test_d1_uniform(dev, orc, bit):
    x = synth.sorted_gaps(10 + bit, NBLK * 128, bit, first=5)
    offs = synth.block_offsets(x, 5)
    wild = synth.rand_u32(11, NBLK * 128)
    woffs = synth.rand_u32(12, NBLK)
    for masked, v, o in ((False, x, offs), (True, x, offs), (True, wild, woffs), (False, wild, woffs)):
        want = orc.batch_pack("d1", v, bit, seeds=o, masked=masked)
        got = to_np(dev.packd1(to_dev(v), bit, initvalues=to_dev(o), masked=masked))
        assert np.array_equal(got, want), ("packd1", bit, masked)
        dec = to_np(dev.unpackd1(to_dev(o), to_dev(want), NBLK, bit))
        assert np.array_equal(dec, orc.batch_unpack("d1", want, NBLK, bit, seeds=o))
    # round trip of sorted data, unmasked, and the chained mode that derives the seeds from the input
    import torch

    seeds_out = torch.empty(NBLK, dtype=torch.int32, device="cuda")
    got = dev.packd1(to_dev(x), bit, initvalue=5, masked=False, initvalues_out=seeds_out)
    assert np.array_equal(to_np(seeds_out), offs)
    assert np.array_equal(to_np(got), orc.batch_pack("d1", x, bit, seeds=offs))
    assert np.array_equal(to_np(dev.unpackd1(seeds_out, got, NBLK, bit)), x)


@pytest.mark.parametrize("bit", BITS)
def test_for_uniform(dev, orc, bit):
    x, base = synth.clustered(20 + bit, NBLK, bit)
    wild = synth.rand_u32(21, NBLK * 128)
    for v in (x, wild):
        want = orc.batch_pack("for", v, bit, seeds=base)
        got = to_np(dev.packFOR(to_dev(base), to_dev(v), bit))
        assert np.array_equal(got, want), ("packFOR", bit)
        dec = to_np(dev.unpackFOR(to_dev(base), to_dev(want), NBLK, bit))
        assert np.array_equal(dec, orc.batch_unpack("for", want, NBLK, bit, seeds=base))
    assert np.array_equal(to_np(dev.unpackFOR(to_dev(base), dev.packFOR(to_dev(base), to_dev(x), bit), NBLK, bit)), x)


@pytest.mark.parametrize("nblk", [0, 1, 2, 7, 8, 31, 32, 33, 63, 64, 65, 257])
def test_ragged_block_counts(dev, orc, nblk):
    for bit in (0, 1, 5, 16, 31, 32):
        x = synth.sorted_gaps(30, nblk * 128, bit, first=9)
        offs = synth.block_offsets(x, 9)
        import torch

        canary = torch.full((nblk * 16 * bit + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        dev.packd1(to_dev(x), bit, initvalues=to_dev(offs) if nblk else None, out=canary)
        got = to_np(canary)
        assert np.array_equal(got[: nblk * 16 * bit], orc.batch_pack("d1", x, bit, seeds=offs))
        assert (got[nblk * 16 * bit:] == 0xA5).all(), "pack wrote past its payload"
        out = torch.full((nblk * 128 + 16,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        dev.unpackd1(to_dev(offs) if nblk else None, canary, nblk, bit, out=out)
        got = to_np(out)
        assert np.array_equal(got[: nblk * 128], x)
        assert (got[nblk * 128:] == 0x5A5A5A5A).all(), "unpack wrote past its output"


def test_long_arrays_narrow_widths(dev, orc):
    """Narrow widths interleave prefetch workgroups with the decoding ones; an
    array long enough to leave the start-up ramp of the prefetch distance, with a
    block count that is no multiple of the group size."""
    nblk = 8 * 16 * 700 + 8 * 5 + 3
    for mode, bit in (("d1", 1), ("d1", 3), ("d1", 7), ("plain", 2), ("for", 5), ("d1", 8)):
        if mode == "d1":
            x = synth.sorted_gaps(70 + bit, nblk * 128, bit, first=3)
            seeds = synth.block_offsets(x, 3)
            payload = dev.packd1(to_dev(x), bit, initvalues=to_dev(seeds))
            dec = dev.unpackd1(to_dev(seeds), payload, nblk, bit)
        elif mode == "plain":
            x, seeds = synth.random_bits(71, nblk * 128, bit), None
            payload = dev.pack(to_dev(x), bit)
            dec = dev.unpack(payload, nblk, bit)
        else:
            x, seeds = synth.clustered(72, nblk, bit)
            payload = dev.packFOR(to_dev(seeds), to_dev(x), bit)
            dec = dev.unpackFOR(to_dev(seeds), payload, nblk, bit)
        want = orc.batch_pack(mode, x, bit, seeds=seeds)
        assert np.array_equal(to_np(payload), want), (mode, bit)
        assert np.array_equal(to_np(dec), x), (mode, bit)


def test_reductions(dev, orc):
    for bit in BITS:
        x = synth.sorted_gaps(40 + bit, NBLK * 128, bit, first=77)
        offs = synth.block_offsets(x, 5)
        r = synth.random_bits(41, NBLK * 128, bit)
        assert np.array_equal(to_np(dev.maxbits(to_dev(r))), orc.batch_maxbits(r))
        assert np.array_equal(to_np(dev.maxbitsd1(to_dev(x), initvalues=to_dev(offs))), orc.batch_maxbitsd1(offs, x))
        assert np.array_equal(to_np(dev.maxbitsd1(to_dev(x), initvalue=5)), orc.batch_maxbitsd1(offs, x))
        assert np.array_equal(to_np(dev.maxbitsd1(to_dev(r), initvalue=5)),
                              orc.batch_maxbitsd1(synth.block_offsets(r, 5), r))
        lo, hi, bits = dev.maxmin(to_dev(r))
        wlo, whi = orc.batch_minmax(r)
        assert np.array_equal(to_np(lo), wlo) and np.array_equal(to_np(hi), whi)
        assert np.array_equal(to_np(bits), np.array([orc.bits(int(h) - int(l)) for l, h in zip(wlo, whi)], np.uint8))
        for n in (0, 1, 2, 3, 4, 5, 127, 128, 129, 1000, NBLK * 128):
            got = to_np(dev.reduce_length(to_dev(r), n, initvalue=9))
            lo_, hi_ = orc.simdmaxmin_length(r, n)
            assert orc.bits(int(got[0])) == orc.maxbits_length(r, n)
            if n:
                assert orc.bits(int(got[1])) == orc.simdmaxbitsd1_length(9, r, n)
            assert (int(got[2]), int(got[3])) == (lo_, hi_)


def test_mixed_widths(dev, orc):
    """The example.c:56-75 caller convention on the device: widths from
    simdmaxbitsd1, payload offsets from the scan, pack, unpack."""
    nblk = 3000
    v, offs = G.corpus_mixed(nblk)
    x = to_dev(v)
    import torch

    seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(x, initvalue=0, initvalues_out=seeds)
    wbits = orc.batch_maxbitsd1(offs, v)
    assert np.array_equal(to_np(bits), wbits) and np.array_equal(to_np(seeds), offs)
    off, total = dev.payload_offsets(bits)
    pos = np.concatenate([[0], np.cumsum(wbits.astype(np.uint64) * 16)]).astype(np.uint64)
    assert np.array_equal(to_np(off), pos[:-1]) and int(total.item()) == int(pos[-1])
    payload = torch.full((int(pos[-1]) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(1, x, bits, off, payload, initvalues=seeds)
    want = orc.batch_pack("d1", v, seeds=offs, bits=wbits, off=pos[:-1].copy(), out_bytes=int(pos[-1]))
    got = to_np(payload)
    assert np.array_equal(got[: int(pos[-1])], want) and (got[int(pos[-1]):] == 0xA5).all()
    dec = dev.unpack_mixed(1, payload, bits, off, nblk, initvalues=seeds)
    assert np.array_equal(to_np(dec), v)
    # plain and FOR with arbitrary per-block widths 0..32 (including 0 and 32) on arbitrary data
    wb = (synth.rand_u32(50, nblk) % 33).astype(np.uint8)
    pos = np.concatenate([[0], np.cumsum(wb.astype(np.uint64) * 16)]).astype(np.uint64)
    wild = synth.rand_u32(51, nblk * 128)
    base = synth.rand_u32(52, nblk)
    for mode, name, masked in ((0, "plain", False), (0, "plain", True), (1, "d1", False), (1, "d1", True),
                               (2, "for", False)):
        want = orc.batch_pack(name, wild, seeds=base, bits=wb, off=pos[:-1].copy(), out_bytes=int(pos[-1]),
                              masked=masked)
        payload = torch.zeros(int(pos[-1]) + 16, dtype=torch.uint8, device="cuda")
        dev.pack_mixed(mode, to_dev(wild), to_dev(wb), to_dev(pos[:-1].copy()), payload, initvalues=to_dev(base),
                       masked=masked)
        assert np.array_equal(to_np(payload)[: int(pos[-1])], want), (name, masked)
        dec = dev.unpack_mixed(mode, to_dev(want), to_dev(wb), to_dev(pos[:-1].copy()), nblk, initvalues=to_dev(base))
        assert np.array_equal(to_np(dec), orc.batch_unpack(name, want, nblk, seeds=base, bits=wb, off=pos[:-1].copy()))


@pytest.mark.parametrize("shift", [1, 2, 3, 5])
def test_mixed_widths_metadata_at_any_alignment(dev, orc, shift):
    """The decoder fetches a wave's widths, offsets and seeds as aligned words around them: width arrays that start
    at any byte, offsets and seeds at their natural alignment only."""
    import torch

    nblk = 3000
    wb = (synth.rand_u32(90 + shift, nblk) % 33).astype(np.uint8)
    pos = np.concatenate([[0], np.cumsum(wb.astype(np.uint64) * 16)]).astype(np.uint64)
    wild = synth.rand_u32(91, nblk * 128)
    base = synth.rand_u32(92, nblk)
    bits_buf = torch.full((nblk + 16,), 77, dtype=torch.uint8, device="cuda")
    bits_buf[shift:shift + nblk] = to_dev(wb)
    off_buf = torch.zeros(nblk + 1, dtype=torch.int64, device="cuda")
    off_buf[1:] = to_dev(pos[:-1].copy())
    seeds_buf = torch.zeros(nblk + 1, dtype=torch.int32, device="cuda")
    seeds_buf[1:] = to_dev(base)
    for mode, name in ((0, "plain"), (1, "d1"), (2, "for")):
        want = orc.batch_pack(name, wild, seeds=base, bits=wb, off=pos[:-1].copy(), out_bytes=int(pos[-1]))
        dec = dev.unpack_mixed(mode, to_dev(want), bits_buf[shift:shift + nblk], off_buf[1:], nblk, initvalues=seeds_buf[1:])
        assert np.array_equal(to_np(dec), orc.batch_unpack(name, want, nblk, seeds=base, bits=wb, off=pos[:-1].copy()))
        payload = torch.zeros(int(pos[-1]) + 16, dtype=torch.uint8, device="cuda")
        dev.pack_mixed(mode, to_dev(wild), bits_buf[shift:shift + nblk], off_buf[1:], payload, initvalues=seeds_buf[1:],
                       masked=False)
        assert np.array_equal(to_np(payload)[: int(pos[-1])], want)


@pytest.mark.parametrize("layout", ["arena", "stream", "reversed", "padded"])
def test_long_narrow_mixed_widths(dev, orc, layout):
    """Groups of narrow blocks have their payload touched ahead of the decoding workgroups, from where the first
    block's payload starts to where the last one's ends: long enough to leave the start-up ramp of that distance,
    with the payloads end to end (aligned arena, byte stream), in descending order, and with gaps between them."""
    import torch

    nblk = 256 * 700 + 45
    wb = np.minimum(synth.rand_u32(80, nblk) % 8, synth.rand_u32(81, nblk) % 13).astype(np.uint8)  # 0..7, mostly small
    wb[5000:5600] = 32  # a stretch of raw blocks: its groups are left alone
    sizes = wb.astype(np.uint64) * 16
    pos = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    total = int(pos[-1])
    gaps = synth.rand_u32(82, nblk * 128) & np.repeat(((np.uint64(1) << wb.astype(np.uint64)) - 1).astype(np.uint32), 128)
    ids = np.cumsum(gaps, dtype=np.uint32)
    seeds = synth.block_offsets(ids, 0)
    if layout == "stream":
        strm = torch.empty(total + nblk, dtype=torch.uint8, device="cuda")
        dev.pack_stream(1, to_dev(ids), to_dev(wb), to_dev(pos[:-1].copy()), strm, initvalue=0)
        dec = dev.unpack_stream(1, strm, to_dev(wb), to_dev(pos[:-1].copy()), nblk, initvalues=to_dev(seeds))
        assert np.array_equal(to_np(dec), ids)
        return
    if layout == "arena":
        off, nbytes = pos[:-1].copy(), total
    elif layout == "reversed":
        off, nbytes = (np.uint64(total) - pos[1:]).astype(np.uint64), total
    else:
        off, nbytes = (pos[:-1] + np.arange(nblk, dtype=np.uint64) * 16).astype(np.uint64), total + 16 * nblk
    want = orc.batch_pack("d1", ids, seeds=seeds, bits=wb, off=off.copy(), out_bytes=nbytes)
    payload = torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(1, to_dev(ids), to_dev(wb), to_dev(off), payload, initvalues=to_dev(seeds))
    assert np.array_equal(to_np(payload)[:nbytes], want)
    dec = dev.unpack_mixed(1, payload, to_dev(wb), to_dev(off), nblk, initvalues=to_dev(seeds))
    assert np.array_equal(to_np(dec), ids)


@pytest.mark.parametrize("bit", BITS)
def test_lengths(dev, orc, bit):
    full = {"inrange": synth.random_bits(60, 20011, bit), "wild": synth.rand_u32(61, 20011)}
    import torch

    for name, data in full.items():
        for n in [0, 1, 2, 3, 4, 5, 7, 8, 31, 127, 128, 129, 130, 255, 256, 257, 1000, 4095, 4097, 5000, 20011]:
            x = data[:n].copy()
            want = orc.simdpack_length(x, bit)
            nbytes = dev.compressedbytes(n, bit)
            assert nbytes == len(want) == orc.compressedbytes(n, bit)
            xd = to_dev(x) if n else torch.zeros(4, dtype=torch.int32, device="cuda")
            buf = torch.full((nbytes + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            dev.pack_length(xd[:n], bit, out=buf)
            got = to_np(buf)
            assert np.array_equal(got[:nbytes], want), ("pack_length", name, n, bit)
            assert (got[nbytes:] == 0xA5).all()
            out = torch.full((n + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            dev.unpack_length(buf, n, bit, out=out)
            wdec, _ = orc.simdunpack_length(want, n, bit)
            got = to_np(out)
            assert np.array_equal(got[:n], wdec) and (got[n:] == 0x5A5A5A5A).all()
            base = 0x00C0FFEE
            want = orc.simdpackFOR_length(base, x, bit)
            buf = torch.full((nbytes + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            dev.packFOR_length(base, xd[:n], bit, out=buf)
            got = to_np(buf)
            assert np.array_equal(got[:nbytes], want), ("packFOR_length", name, n, bit)
            assert (got[nbytes:] == 0xA5).all()
            out = torch.full((n + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            dev.unpackFOR_length(base, buf, n, bit, out=out)
            wdec, _ = orc.simdunpackFOR_length(base, want, n, bit)
            got = to_np(out)
            assert np.array_equal(got[:n], wdec) and (got[n:] == 0x5A5A5A5A).all()


def test_golden_blocks(dev):
    """Device kernels against the fixtures produced by the compiled reference."""
    g = G.load("blocks.npz")
    import torch

    for bit in BITS:
        for kind in range(3):
            x = to_dev(g["x"][bit, kind].copy())
            iv = np.array([g["init"][bit, kind]], np.uint32)
            outs = [dev.pack(x, bit, masked=True), dev.pack(x, bit, masked=False),
                    dev.packd1(x, bit, initvalues=to_dev(iv), masked=True),
                    dev.packd1(x, bit, initvalues=to_dev(iv), masked=False), dev.packFOR(to_dev(iv), x, bit)]
            for j, o in enumerate(outs):
                want = g["packed"][bit, kind, j, :16 * bit]
                assert np.array_equal(to_np(o), want), (G.PACK_NAMES[j], bit, kind)
                wd = to_dev(want.copy()) if bit else torch.zeros(16, dtype=torch.uint8, device="cuda")
                assert np.array_equal(to_np(dev.unpack(wd, 1, bit)), g["unpacked"][bit, kind, j, 0])
                assert np.array_equal(to_np(dev.unpackd1(to_dev(iv), wd, 1, bit)), g["unpacked"][bit, kind, j, 1])
                assert np.array_equal(to_np(dev.unpackFOR(to_dev(iv), wd, 1, bit)), g["unpacked"][bit, kind, j, 2])
            lo, hi, _ = dev.maxmin(x)
            red = [int(to_np(dev.maxbits(x))[0]), int(to_np(dev.maxbitsd1(x, initvalues=to_dev(iv)))[0]),
                   int(to_np(lo)[0]), int(to_np(hi)[0])]
            assert red == g["red"][bit, kind].tolist()


def test_golden_corpus_digests(dev, orc):
    g = G.load("corpus.npz")
    nblk = int(g["nblocks"])
    for mode, bit, masked, hpay, hdec in g["table"].tolist():
        if mode == 0:
            v = G.corpus_plain(bit, masked, nblk)
            pay = dev.pack(to_dev(v), bit, masked=bool(masked))
            dec = dev.unpack(pay, nblk, bit)
        elif mode == 1:
            v, offs = G.corpus_d1(bit, nblk)
            pay = dev.packd1(to_dev(v), bit, initvalues=to_dev(offs), masked=bool(masked))
            dec = dev.unpackd1(to_dev(offs), pay, nblk, bit)
        else:
            v, base = G.corpus_for(bit, nblk)
            pay = dev.packFOR(to_dev(base), to_dev(v), bit)
            dec = dev.unpackFOR(to_dev(base), pay, nblk, bit)
        assert orc.fnv1a64(to_np(pay)) == hpay, (mode, bit, masked)
        assert orc.fnv1a64(to_np(dec)) == hdec, (mode, bit, masked)


def test_argument_errors(dev):
    import torch

    from simdcomp_amd import binding as B

    x = torch.zeros(256, dtype=torch.int32, device="cuda")
    out = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    L = B.lib()
    assert L.simdcomp_hip_pack(x.data_ptr(), out.data_ptr(), 2, 33, 0, None) == -1
    assert L.simdcomp_hip_pack(x.data_ptr() + 4, out.data_ptr(), 1, 5, 0, None) == -2
    assert L.simdcomp_hip_pack(None, out.data_ptr(), 1, 5, 0, None) == -3
    assert L.simdcomp_hip_unpack_mixed(7, None, out.data_ptr(), None, None, x.data_ptr(), 1, None) == -4
    assert L.simdcomp_hip_unpackd1(None, out.data_ptr(), x.data_ptr(), 1, 5, None) == -3
    assert b"NULL" in L.simdcomp_hip_last_error()
    assert L.simdcomp_hip_pack(None, None, 0, 5, 0, None) == 0  # empty batch is fine
    # the directory forms of the queries
    d, o, q = out.data_ptr(), x.data_ptr(), x.data_ptr()
    assert L.simdcomp_hip_block_directory(0, None, None, d, 4, 0, d, None) == -3
    assert L.simdcomp_hip_block_directory(0, None, o, d, 4, 0, d + 8, None) == -2
    assert L.simdcomp_hip_block_directory(0, None, None, None, 0, 0, None, None) == 0
    assert L.simdcomp_hip_block_directory_bytes(10) == 160
    assert L.simdcomp_hip_select_directory(9, d, d, None, q, 4, q, None) == -4
    assert L.simdcomp_hip_select_directory(1, d, None, None, q, 4, q, None) == -3
    assert L.simdcomp_hip_select_directory(1, d, d + 4, None, q, 4, q, None) == -2
    assert L.simdcomp_hip_search_directory(2, d, d, None, q, None, 4, q, q, q, None) == -4  # the offset vector is d1's
    assert L.simdcomp_hip_search_directory(1, d, d, None, None, None, 4, q, q, None, None) == -3
    assert L.simdcomp_hip_search_directory(1, None, None, None, None, None, 0, None, None, None, None) == 0


def test_partition_and_checksum(dev):
    import torch

    total = 1000003
    covered = 0
    for nparts in (1, 2, 4, 8):
        prev_end = 0
        for p in range(nparts):
            first, count = dev.partition(total, nparts, p)
            assert first == prev_end
            prev_end = first + count
        assert prev_end == total
    x = to_dev(synth.rand_u32(70, 128 * 1000))
    whole = dev.checksum(x)
    acc = torch.zeros(2, dtype=torch.int64, device="cuda")
    for p in range(4):
        first, count = dev.partition(1000, 4, p)
        dev.checksum(x[first * 128:(first + count) * 128], index0=first * 128, acc=acc)
    assert acc[0].item() == whole[0].item()
    y = x.clone()
    y[5] += 1
    y[99999] ^= 4
    assert dev.checksum(x, other=y)[1].item() == 2
    assert dev.checksum(y)[0].item() != whole[0].item()


@pytest.mark.parametrize("seed", range(6))
def test_randomised_shapes(dev, orc, seed):
    """Random block counts (around the workgroup, prefetch-group and launch granularities), modes, widths and
    seed conventions against the oracle."""
    import torch

    rng = np.random.default_rng(1000 + seed)
    for _ in range(12):
        nblk = int(rng.choice([1, 3, 8, 9, 127, 128, 129, 136, 137, 2047, 2048, 2177, 5000, 20001]))
        bit = int(rng.integers(0, 33))
        mode = ["plain", "d1", "for"][int(rng.integers(0, 3))]
        masked = bool(rng.integers(0, 2)) and mode != "for"
        wild = bool(rng.integers(0, 2))
        x = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk * 128)
        if not wild:
            x = x & synth.mask(bit) if bit < 32 else x
            if mode == "d1":
                x = (np.cumsum(x.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        seeds = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk) if mode != "plain" else None
        if mode == "d1" and bool(rng.integers(0, 2)):
            seeds = synth.block_offsets(x, 11)  # chained from the input
            got = dev.packd1(to_dev(x), bit, initvalue=11, masked=masked)
        elif mode == "d1":
            got = dev.packd1(to_dev(x), bit, initvalues=to_dev(seeds), masked=masked)
        elif mode == "for":
            got = dev.packFOR(to_dev(seeds), to_dev(x), bit)
        else:
            got = dev.pack(to_dev(x), bit, masked=masked)
        want = orc.batch_pack(mode, x, bit, seeds=seeds, masked=masked)
        assert np.array_equal(to_np(got), want), (mode, bit, nblk, masked, wild)
        if mode == "d1":
            dec = dev.unpackd1(to_dev(seeds), got, nblk, bit)
        elif mode == "for":
            dec = dev.unpackFOR(to_dev(seeds), got, nblk, bit)
        else:
            dec = dev.unpack(got, nblk, bit)
        assert np.array_equal(to_np(dec), orc.batch_unpack(mode, want, nblk, bit, seeds=seeds)), (mode, bit, nblk)


def test_reduce_length_exact_at_every_boundary(dev):
    """The whole-array reduction against numpy, exactly (not only the widths of its results): arrays that are one
    step up at position i have a single non-zero difference, which any wrong predecessor at a lane / step / wave /
    workgroup / kernel boundary would add bits to; sizes around the bulk kernel's workgroups, aligned and not."""
    import torch

    rng = np.random.default_rng(77)
    per_wg = 8192
    for n in (1, 5, 255, 256, 257, 2047, 2048, 2049, per_wg - 1, per_wg, per_wg + 1, 3 * per_wg + 77, 5 * per_wg):
        marks = sorted({0, 1, 3, 4, 255, 256, 257, 2047, 2048, per_wg - 1, per_wg, per_wg + 1, 2 * per_wg, n - 1, n // 2})
        for i in [m for m in marks if m < n]:
            c, step = 1000 + i, 1 << int(rng.integers(0, 31))
            x = np.full(n + 8, c, dtype=np.uint32)
            x[i:] = (c + step) & 0xFFFFFFFF
            for shift in (0, 1):  # 16-byte aligned, and only 4-byte aligned
                buf = torch.from_numpy(np.concatenate([np.zeros(shift, np.uint32), x]).view(np.int32)).cuda()
                got = to_np(dev.reduce_length(buf[shift:], n, initvalue=c))
                want_or = int(np.bitwise_or.reduce(x[:n]))
                assert int(got[0]) == want_or, (n, i, shift)
                assert int(got[1]) == step, (n, i, shift, hex(int(got[1])))
                assert (int(got[2]), int(got[3])) == (int(x[:n].min()), int(x[:n].max())), (n, i, shift)
    # random data, exact
    for n in (per_wg * 4 + 1234, 100001):
        x = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        got = to_np(dev.reduce_length(to_dev(x), n, initvalue=7))
        d = np.diff(np.concatenate([[np.uint32(7)], x]).astype(np.uint32))
        assert int(got[0]) == int(np.bitwise_or.reduce(x)) and int(got[1]) == int(np.bitwise_or.reduce(d))
        assert (int(got[2]), int(got[3])) == (int(x.min()), int(x.max()))
