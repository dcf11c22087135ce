"""Device-side handling of the example.c byte stream ([width byte][payload] per
block, example.c:56-75) and of chained differential decoding without stored
offsets (example.c:101-110), through the C ABI, against the CPU loops run over
the oracle and against the streams the reference itself wrote (tests/golden)."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

D1, PLAIN = 1, 0


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


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def compress_on_device(dev, x, d1=True):
    """example.c:56-75 on the device: widths, offsets scan, pack, stream build."""
    import torch

    d_x = to_dev(x)
    seeds = torch.empty(x.size // 128, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(d_x, initvalue=0, initvalues_out=seeds) if d1 else dev.maxbits(d_x)
    off, total = dev.payload_offsets(bits)
    nbytes = int(total.item())
    arena = torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(D1 if d1 else PLAIN, d_x, bits, off, arena, initvalues=seeds if d1 else None)
    stream = dev.stream_build(arena, bits, off, nbytes)
    return stream, arena, bits, off, seeds, nbytes


@pytest.mark.parametrize("d1", [True, False])
def test_golden_stream_round_trip(dev, d1):
    """The device writes byte for byte the stream the reference wrote, and reads it back."""
    g = G.load("stream.npz")
    nblk = int(g["nblocks"])
    want = g["d1" if d1 else "plain"]
    x = G.stream_corpus(900, nblk)
    stream, arena, bits, off, seeds, nbytes = compress_on_device(dev, x, d1)
    assert np.array_equal(stream.cpu().numpy(), want)
    # decode from the bytes alone: index -> re-aligned arena -> (chained offsets) -> unpack
    ibits, ioff, res = dev.stream_index(to_dev(want), nblk)
    assert res.tolist() == [nblk, want.size - nblk, 0]
    assert np.array_equal(ibits.cpu().numpy(), bits.cpu().numpy())
    assert np.array_equal(ioff.cpu().numpy(), off.cpu().numpy())
    arena2 = dev.stream_unbuild(to_dev(want), ibits, ioff, nbytes)
    assert np.array_equal(arena2.cpu().numpy()[:nbytes], arena.cpu().numpy()[:nbytes])
    if d1:
        chained = dev.chain_offsets(arena2, ibits, ioff, initvalue=0)
        assert np.array_equal(u32(chained), u32(seeds))
        out = dev.unpack_mixed(D1, arena2, ibits, ioff, nblk, initvalues=chained)
    else:
        out = dev.unpack_mixed(PLAIN, arena2, ibits, ioff, nblk)
    assert np.array_equal(u32(out), x)
    # the same without the aligned arena: pack straight into a stream, decode straight from its bytes
    import torch

    direct = torch.full((want.size + 3,), 0xEE, dtype=torch.uint8, device="cuda")
    dev.pack_stream(D1 if d1 else PLAIN, to_dev(x), bits, off, direct[3:], initvalues=seeds if d1 else None)
    assert np.array_equal(direct[3:].cpu().numpy(), want) and (direct[:3] == 0xEE).all()
    if d1:
        chained = dev.chain_offsets(direct[3:], ibits, ioff, initvalue=0, stream_layout=True)
        assert np.array_equal(u32(chained), u32(seeds))
        out = dev.unpack_stream(D1, direct[3:], ibits, ioff, nblk, initvalues=chained)
    else:
        out = dev.unpack_stream(PLAIN, direct[3:], ibits, ioff, nblk)
    assert np.array_equal(u32(out), x)


@pytest.mark.parametrize("nblk", [1, 2, 9, 130, 4097, 70001])
def test_stream_sizes_and_alignments(dev, orc, nblk):
    """Record counts around the scan tiles and index chunks; the stream buffer at every byte alignment."""
    import torch

    x = G.stream_corpus(40 + nblk, nblk)
    want = orc.stream_compress(x, d1=True, initvalue=0)
    stream, arena, bits, off, seeds, nbytes = compress_on_device(dev, x)
    assert np.array_equal(stream.cpu().numpy(), want)
    n, wbits, woff, status = orc.stream_walk(want, nblk)
    assert (n, status) == (nblk, 0)
    for shift in (1, 2, 3, 5):
        buf = torch.full((want.size + 16,), 0xEE, dtype=torch.uint8, device="cuda")
        view = buf[shift:shift + want.size]
        dev.stream_build(arena, bits, off, nbytes, out=view)
        got = buf.cpu().numpy()
        assert np.array_equal(got[shift:shift + want.size], want)
        assert (got[:shift] == 0xEE).all() and (got[shift + want.size:] == 0xEE).all(), "wrote outside the stream"
        ibits, ioff, res = dev.stream_index(view, nblk + 5)
        assert res.tolist() == [nblk, want.size - nblk, 0]
        assert np.array_equal(ibits.cpu().numpy()[:nblk], wbits)
        assert np.array_equal(ioff.cpu().numpy()[:nblk].view(np.uint64), woff)
        arena2 = torch.full((nbytes + 32,), 0xCC, dtype=torch.uint8, device="cuda")
        dev.stream_unbuild(view, ibits[:nblk], ioff[:nblk], nbytes, out=arena2)
        got = arena2.cpu().numpy()
        assert np.array_equal(got[:nbytes], arena.cpu().numpy()[:nbytes])
        assert (got[nbytes:] == 0xCC).all(), "wrote outside the arena"
        # in place: pack_stream (chained from the input, like example.c:63-71) and unpack_stream at this alignment
        buf2 = torch.full((want.size + 16,), 0xDD, dtype=torch.uint8, device="cuda")
        view2 = buf2[shift:shift + want.size]
        dev.pack_stream(D1, to_dev(x), bits, off, view2, initvalue=0)
        got = buf2.cpu().numpy()
        assert np.array_equal(got[shift:shift + want.size], want)
        assert (got[:shift] == 0xDD).all() and (got[shift + want.size:] == 0xDD).all(), "wrote outside the stream"
        chained = dev.chain_offsets(view2, bits, off, initvalue=0, stream_layout=True)
        assert np.array_equal(u32(chained), u32(seeds))
        out = torch.full((nblk * 128 + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        dev.unpack_stream(D1, view2, bits, off, nblk, initvalues=chained, out=out)
        assert np.array_equal(u32(out[:nblk * 128]), x) and (out[nblk * 128:] == 0x5A5A5A5A).all()
    for init in (0, 0xDEADBEEF):
        # offsets chained from another start value: decode equals the CPU loop of example.c:101-110
        chained = dev.chain_offsets(arena, bits, off, initvalue=init)
        out = dev.unpack_mixed(D1, arena, bits, off, nblk, initvalues=chained)
        assert np.array_equal(u32(out), orc.stream_decode(want, nblk, d1=True, initvalue=init))


@pytest.mark.parametrize("layout", ["reversed", "padded", "far"])
def test_chain_offsets_with_any_payload_offsets(dev, orc, layout):
    """Payload offsets need not come from a scan: any multiples of 16 are legal for pack_mixed / unpack_mixed, so
    chain_offsets must take them too -- blocks laid out back to front (negative distances inside a wave), padded
    apart, and more than 4 GiB apart (distances that do not fit 32 bits)."""
    import torch

    nblk = 1000 + 7
    x = G.stream_corpus(321, nblk)
    d_x = to_dev(x)
    seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(d_x, initvalue=0, initvalues_out=seeds)
    scan, total = dev.payload_offsets(bits)
    nbytes = int(total.item())
    size = 16 * bits.to(torch.int64)
    if layout == "reversed":
        off = nbytes - scan - size  # block 0 last
        span = nbytes
    elif layout == "padded":
        off = scan + 48 * torch.arange(nblk, device="cuda")
        span = nbytes + 48 * nblk
    else:
        if torch.cuda.mem_get_info()[0] < 12 * (1 << 30):
            pytest.skip("needs 10 GiB of free HBM")
        far = (torch.arange(nblk, device="cuda") % 3 == 1).to(torch.int64) * (5 << 30)  # every third block 5 GiB up
        off = scan + far
        span = nbytes + (5 << 30)
    arena = torch.empty(span + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(D1, d_x, bits, off, arena, initvalues=seeds)
    chained = dev.chain_offsets(arena, bits, off, initvalue=0)
    assert np.array_equal(u32(chained), u32(seeds))
    out = dev.unpack_mixed(D1, arena, bits, off, nblk, initvalues=chained)
    assert np.array_equal(u32(out), x)
    del arena


def test_index_rejects_bad_streams(dev, orc):
    x = G.stream_corpus(77, 3000)
    s = orc.stream_compress(x, d1=True)
    d = to_dev(s)
    assert dev.stream_index(d, 3000)[2].tolist()[2] == 0
    assert dev.stream_index(d[:-1], 3000)[2].tolist()[2] == 1  # ends inside a record
    n, bits, off, _ = orc.stream_walk(s, 3000)
    for k in (0, 1500, 2999):
        bad = s.copy()
        bad[int(off[k]) + k] = 33 + k % 200
        assert dev.stream_index(to_dev(bad), 3000)[2].tolist()[2] == 1
    # payload bytes above 32 are not width bytes and do not matter
    assert (s > 32).any()
    # empty stream
    import torch

    assert dev.stream_index(torch.empty(0, dtype=torch.uint8, device="cuda"), 4)[2].tolist() == [0, 0, 0]


@pytest.mark.parametrize("d1", [True, False])
def test_decode_of_truncated_streams_reads_no_record_past_the_end(dev, orc, d1):
    """simdcomp_hip_stream_decode on streams that end inside their last record (one byte short; a last width byte of
    32 with nothing behind it): reported in d_result[2], the cut record is not decoded (no kernel reads a payload that
    lies beyond the stream), and the records in front decode as before."""
    import torch

    from simdcomp_amd import binding as B

    L = B.lib()
    nblk = 700
    x = G.stream_corpus(55, nblk)
    s = orc.stream_compress(x, d1=d1)
    n, bits, off, _ = orc.stream_walk(s, nblk)
    last = int(off[nblk - 1]) + nblk - 1  # position of the last record's width byte
    want = orc.stream_decode(s, nblk, d1=d1)
    cases = [s[:-1]] if bits[nblk - 1] else []
    cases.append(np.concatenate([s[:last], [32]]).astype(np.uint8))           # width byte 32, no payload at all
    cases.append(np.concatenate([s[:last], [32], np.full(511, 0xEE)]).astype(np.uint8))  # one byte short of a 32-bit record
    for bad in cases:
        T = bad.size
        # the stream ends exactly at the end of its buffer; what follows in memory is another tensor full of 0xFF
        d = to_dev(bad)
        out = torch.zeros(nblk * 128, dtype=torch.int32, device="cuda")
        res = torch.zeros(3, dtype=torch.int64, device="cuda")
        ws = torch.full((max(16, L.simdcomp_hip_stream_decode_workspace(T, nblk)),), 0xFF, dtype=torch.uint8, device="cuda")
        B.check(L.simdcomp_hip_stream_decode(D1 if d1 else PLAIN, 0, B.ptr(d), T, nblk, B.ptr(out), B.ptr(res), B.ptr(ws),
                                             torch.cuda.current_stream().cuda_stream), "stream_decode")
        torch.cuda.synchronize()
        assert res.tolist()[2] == 1
        assert np.array_equal(u32(out)[: (nblk - 1) * 128], want[: (nblk - 1) * 128])


def test_index_dense_and_sparse_records(dev, orc):
    """All-zero widths (one record per byte: tens of thousands of records per index chunk) and all-32 widths."""
    import torch

    n0 = 200001
    z = torch.zeros(n0, dtype=torch.uint8, device="cuda")
    bits, off, res = dev.stream_index(z, n0)
    assert res.tolist() == [n0, 0, 0]
    assert int(bits.max().item()) == 0 and int(off.max().item()) == 0
    nblk = 3001
    x = (np.arange(nblk * 128, dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(0x5A5A5A5A)
    s = orc.stream_compress(x, d1=True)
    assert s.size == nblk * 513
    bits, off, res = dev.stream_index(to_dev(s), nblk)
    assert res.tolist() == [nblk, nblk * 512, 0]
    assert (bits == 32).all() and np.array_equal(off.cpu().numpy(), np.arange(nblk, dtype=np.int64) * 512)


def test_large_stream_2p27(dev):
    """2^27 integers: device compress -> stream -> index -> unbuild -> chain -> decode equals the input."""
    import torch

    nblk = (1 << 27) // 128
    w = torch.minimum(torch.randint(0, 21, (nblk,), device="cuda"), torch.randint(0, 21, (nblk,), device="cuda"))
    gaps = torch.randint(0, 1 << 30, (nblk * 128,), dtype=torch.int32, device="cuda")
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    wild = torch.randint(0, nblk, (nblk // 50,), device="cuda")
    ids.view(-1, 128)[wild] = torch.randint(-2 ** 31, 2 ** 31 - 1, (wild.numel(), 128), dtype=torch.int32, device="cuda")
    seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    assert int((bits == 32).sum().item()) > 0 and int((bits == 0).sum().item()) > 0
    off, total = dev.payload_offsets(bits)
    nbytes = int(total.item())
    arena = torch.empty(nbytes + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(D1, ids, bits, off, arena, initvalues=seeds)
    stream = dev.stream_build(arena, bits, off, nbytes)
    ibits, ioff, res = dev.stream_index(stream, nblk)
    assert res.tolist() == [nblk, nbytes, 0]
    assert torch.equal(ibits, bits) and torch.equal(ioff, off)
    arena2 = dev.stream_unbuild(stream, ibits, ioff, nbytes)
    assert torch.equal(arena2[:nbytes], arena[:nbytes])
    chained = dev.chain_offsets(arena2, ibits, ioff, initvalue=0)
    assert torch.equal(chained, seeds)
    out = dev.unpack_mixed(D1, arena2, ibits, ioff, nblk, initvalues=chained)
    assert int(dev.checksum(out, other=ids)[1].item()) == 0
    # and in place, with no arena at all
    stream2 = torch.empty_like(stream)
    dev.pack_stream(D1, ids, bits, off, stream2, initvalue=0)
    assert torch.equal(stream2, stream)
    chained = dev.chain_offsets(stream2, ibits, ioff, initvalue=0, stream_layout=True)
    assert torch.equal(chained, seeds)
    out = dev.unpack_stream(D1, stream2, ibits, ioff, nblk, initvalues=chained)
    assert int(dev.checksum(out, other=ids)[1].item()) == 0


@pytest.mark.parametrize("d1", [True, False])
def test_one_call_compress_and_decode(dev, orc, d1):
    """simdcomp_hip_stream_compress / _decode: example.c's compress() and decode loop, one call each."""
    import torch

    g = G.load("stream.npz")
    nblk = int(g["nblocks"])
    want = g["d1" if d1 else "plain"]
    x = G.stream_corpus(900, nblk)
    mode = D1 if d1 else PLAIN
    buf, nbytes = dev.stream_compress(mode, to_dev(x))
    n = int(nbytes.item())
    assert n == want.size and np.array_equal(buf[:n].cpu().numpy(), want)
    out, res = dev.stream_decode(mode, to_dev(want), nblk)
    assert res.tolist() == [nblk, want.size - nblk, 0] and np.array_equal(u32(out), x)
    # another start offset, a larger corpus, a stream that sits at an odd address
    for nb, init in ((1, 5), (4099, 0x80000001)):
        x = G.stream_corpus(31 + nb, nb)
        want = orc.stream_compress(x, d1=d1, initvalue=init)
        buf, nbytes = dev.stream_compress(mode, to_dev(x), initvalue=init)
        assert int(nbytes.item()) == want.size and np.array_equal(buf[:want.size].cpu().numpy(), want)
        shifted = torch.zeros(want.size + 7, dtype=torch.uint8, device="cuda")
        shifted[7:] = to_dev(want)
        out, res = dev.stream_decode(mode, shifted[7:], nb, initvalue=init)
        assert res.tolist() == [nb, want.size - nb, 0]
        assert np.array_equal(u32(out), orc.stream_decode(want, nb, d1=d1, initvalue=init))
    # asking for more blocks than the stream holds: reported, the surplus decodes as zero-width blocks
    out, res = dev.stream_decode(mode, to_dev(want), nb + 3, initvalue=init)
    assert res.tolist()[0] == nb and np.array_equal(u32(out)[: nb * 128], orc.stream_decode(want, nb, d1=d1, initvalue=init))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_index_adversarial_payloads(dev, orc, seed):
    """Streams built byte by byte: payload bytes that look like width bytes (half of them are <= 32, many are 0)
    start false chains everywhere; lengths chosen so that records end exactly on, just before and just after the
    16 KiB chunk and 4 MiB group boundaries of the index."""
    import torch

    rng = np.random.default_rng(seed)
    for target in (16384 - 1, 16384, 16384 + 1, 3 * 16384, 256 * 16384 - 7, 256 * 16384, 256 * 16384 + 513, 2 * 256 * 16384 + 5):
        widths = []
        total = 0
        while True:
            w = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 13, 20, 31, 32]))
            if total + 1 + 16 * w > target:
                break
            widths.append(w)
            total += 1 + 16 * w
        while total < target:  # top up with zero-width records so that the stream is exactly `target` bytes long
            widths.append(0)
            total += 1
        widths = np.array(widths, np.uint8)
        nblk = widths.size
        pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
        stream = rng.integers(0, 256, target, dtype=np.uint8)
        low = rng.random(target) < 0.5
        stream[low] = rng.integers(0, 33, int(low.sum()), dtype=np.uint8)
        stream[rng.random(target) < 0.2] = 0
        stream[pos] = widths
        n, wbits, woff, status = orc.stream_walk(stream, nblk)
        assert (n, status) == (nblk, 0) and np.array_equal(wbits, widths)
        bits, off, res = dev.stream_index(to_dev(stream), nblk)
        assert res.tolist() == [nblk, target - nblk, 0], target
        assert np.array_equal(bits.cpu().numpy(), widths), target
        assert np.array_equal(off.cpu().numpy().view(np.uint64), woff), target
        # decode in place (plain): equals the CPU loop over the same bytes
        if nblk <= 40000:
            out = dev.unpack_stream(PLAIN, to_dev(np.concatenate([stream, np.zeros(16, np.uint8)]))[:target], bits, off, nblk)
            assert np.array_equal(u32(out), orc.stream_decode(stream, nblk, d1=False))


def decode_call(d1, d, nblk, init=0):
    """simdcomp_hip_stream_decode through the C ABI on a device byte tensor; returns (integers, result[3])."""
    import torch

    from simdcomp_amd import binding as B

    L = B.lib()
    T = d.numel()
    out = torch.full((nblk * 128 + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    ws = torch.full((max(16, L.simdcomp_hip_stream_decode_workspace(T, nblk)),), 0xFF, dtype=torch.uint8, device="cuda")
    B.check(L.simdcomp_hip_stream_decode(D1 if d1 else PLAIN, init & 0xFFFFFFFF, B.ptr(d), T, nblk,
                                         B.ptr(out), B.ptr(res), B.ptr(ws), torch.cuda.current_stream().cuda_stream), "stream_decode")
    torch.cuda.synchronize()
    assert (out[nblk * 128:] == 0x5A5A5A5A).all(), "wrote past the last block"
    return u32(out[: nblk * 128]), res.tolist()


@pytest.mark.parametrize("width", [1, 2, 5])
def test_index_every_byte_starts_a_chain(dev, orc, width):
    """A stream of one repeated byte value v (records of 1 + 16 v bytes whose payload bytes all equal v): a walk from
    ANY byte is valid and never meets the true chain, so every index chunk has 1 + 16 v live chains that differ only in
    phase -- more than the index keeps slots or logs for.  The index has to come out right all the same, and the
    decode equals the CPU loop."""
    import torch

    nblk = 3 * (1 << 20) // (1 + 16 * width)  # about 3 MiB: some 190 index chunks
    nbytes = nblk * (1 + 16 * width)
    stream = np.full(nbytes, width, np.uint8)
    bits, off, res = dev.stream_index(to_dev(stream), nblk)
    assert res.tolist() == [nblk, nbytes - nblk, 0]
    assert (bits == width).all()
    assert np.array_equal(off.cpu().numpy(), np.arange(nblk, dtype=np.int64) * 16 * width)
    # one byte short: the last record does not fit, everything in front of it is indexed and the stream reported bad
    bits, off, res = dev.stream_index(to_dev(stream[:-1]), nblk)
    assert res.tolist()[2] == 1
    assert (bits[: nblk - 1] == width).all()
    assert np.array_equal(off.cpu().numpy()[: nblk - 1], np.arange(nblk - 1, dtype=np.int64) * 16 * width)
    few = 4000
    out, res = decode_call(True, to_dev(stream[: few * (1 + 16 * width)]), few, init=7)
    assert res == [few, few * 16 * width, 0]
    assert np.array_equal(out, orc.stream_decode(stream[: few * (1 + 16 * width)], few, d1=True, initvalue=7))


@pytest.mark.parametrize("seed", [4, 5])
def test_decode_adversarial_payloads(dev, orc, seed):
    """The one-call decode over streams whose payload bytes look like width bytes (false chains everywhere, several
    chains surviving whole chunks, dense runs of zero-width records), lengths around the 2 KiB sub-chunk, 16 KiB chunk
    and 4 MiB group boundaries: differential and plain, against the CPU loop of example.c over the oracle."""
    rng = np.random.default_rng(seed)
    for target in (1, 17, 2047, 2048, 2049, 16384 - 1, 16384, 16384 + 1, 5 * 16384 + 100, 256 * 16384 - 7, 256 * 16384 + 513):
        widths = []
        total = 0
        while True:
            w = int(rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 13, 20, 31, 32]))
            if total + 1 + 16 * w > target:
                break
            widths.append(w)
            total += 1 + 16 * w
        while total < target:
            widths.append(0)
            total += 1
        widths = np.array(widths, np.uint8)
        nblk = widths.size
        pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
        stream = rng.integers(0, 256, target, dtype=np.uint8)
        low = rng.random(target) < 0.5
        stream[low] = rng.integers(0, 33, int(low.sum()), dtype=np.uint8)
        stream[rng.random(target) < 0.2] = 0
        stream[pos] = widths
        d = to_dev(stream)
        for d1 in (True, False):
            for init in (0, 0xFFFFFF00):
                got, res = decode_call(d1, d, nblk, init)
                assert res == [nblk, target - nblk, 0], (target, d1)
                want = orc.stream_decode(stream, nblk, d1=d1, initvalue=init)
                assert np.array_equal(got, want), (target, d1, init)


def test_decode_dense_and_wide_records(dev, orc):
    """All-zero widths (a record per byte, 16384 records per chunk, every block a copy of the start value), all-32
    widths (raw blocks, the chain restarts at each), and long runs of each in one stream."""
    n0 = 70001
    z = to_dev(np.zeros(n0, np.uint8))
    got, res = decode_call(True, z, n0, 0xABCD1234)
    assert res == [n0, 0, 0] and (got == 0xABCD1234).all()
    got, res = decode_call(False, z, n0)
    assert res == [n0, 0, 0] and (got == 0).all()
    nblk = 1500
    x = (np.arange(nblk * 128, dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(0x5A5A5A5A)
    s32 = orc.stream_compress(x, d1=True)
    assert s32.size == nblk * 513
    got, res = decode_call(True, to_dev(s32), nblk)
    assert res == [nblk, nblk * 512, 0] and np.array_equal(got, x)
    # runs: 40000 zero-width records, 300 raw blocks, 40000 zero-width records, a few narrow blocks
    y = G.stream_corpus(5, 64)
    tail = orc.stream_compress(y, d1=True, initvalue=int(x[300 * 128 - 1]))
    mixed = np.concatenate([np.zeros(40000, np.uint8), s32[: 300 * 513], np.zeros(40000, np.uint8), tail])
    n = 80000 + 300 + 64
    for d1 in (True, False):
        got, res = decode_call(d1, to_dev(mixed), n, 77)
        assert res[0] == n and res[2] == 0
        assert np.array_equal(got, orc.stream_decode(mixed, n, d1=d1, initvalue=77))


@pytest.mark.parametrize("d1", [True, False])
def test_decode_large_stream_every_alignment(dev, d1):
    """2^26 integers (several thousand chunks: the chain crosses many look-back windows), the stream at odd addresses,
    fewer and more blocks asked for than the stream holds."""
    import torch

    nblk = (1 << 26) // 128
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    w = torch.minimum(torch.randint(0, 25, (nblk,), device="cuda", generator=g), torch.randint(0, 25, (nblk,), device="cuda", generator=g))
    gaps = torch.randint(0, 1 << 30, (nblk * 128,), dtype=torch.int32, device="cuda", generator=g)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    wild = torch.randint(0, nblk, (nblk // 40,), device="cuda", generator=g)
    ids.view(-1, 128)[wild] = torch.randint(-2 ** 31, 2 ** 31 - 1, (wild.numel(), 128), dtype=torch.int32, device="cuda", generator=g)
    mode = D1 if d1 else PLAIN
    buf, nbytes = dev.stream_compress(mode, ids)
    T = int(nbytes.item())
    for shift in (0, 1, 7):
        shifted = torch.empty(T + 16, dtype=torch.uint8, device="cuda")
        shifted[shift:shift + T] = buf[:T]
        out, res = dev.stream_decode(mode, shifted[shift:shift + T], nblk)
        assert res.tolist() == [nblk, T - nblk, 0]
        assert torch.equal(out, ids)
    out, res = dev.stream_decode(mode, buf[:T], nblk - 1000)
    assert res.tolist()[0] == nblk and torch.equal(out, ids[: (nblk - 1000) * 128])
    out, res = dev.stream_decode(mode, buf[:T], nblk + 50)
    assert res.tolist()[0] == nblk and torch.equal(out[: nblk * 128], ids)
    surplus = out[nblk * 128:]
    assert (surplus == (ids[-1] if d1 else 0)).all()


def test_index_soak(dev, orc):
    """Eighty synthetic streams of many shapes (natural, narrow, wide, zero-width runs; payloads that are random, look
    like width bytes, repeat the record's own width, or are one constant; a third with a bad width byte, a third cut
    short), each compared with the CPU walk: status, record count, and width / offset of every record in front of the
    first bad byte (tools/exp/index_soak.py is the long form of this test)."""
    import torch

    rng = np.random.default_rng(20260)
    menus = ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 17, 20, 24, 32], [0, 0, 1, 1, 1, 2, 3], [28, 30, 31, 32, 32],
             [0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 1, 2, 3, 5, 8, 13, 20, 31, 32])
    for it in range(80):
        target = int(rng.choice([17, 513, 1539, 16384, 16385, 40000, 3 * 16384 - 1, 300000, 1 << 20]))
        if it % 2:
            target = int(rng.integers(1, target + 1))
        widths, total = [], 0
        while True:
            w = int(rng.choice(menus[it % 5]))
            if total + 1 + 16 * w > target:
                break
            widths.append(w)
            total += 1 + 16 * w
        widths = np.array(widths + [0] * (target - total), np.uint8)
        pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
        fill = it % 4
        if fill == 0:
            stream = rng.integers(0, 256, target, dtype=np.uint8)
        elif fill == 1:
            stream = rng.integers(0, 33, target, dtype=np.uint8)
        elif fill == 2:
            stream = np.repeat(widths, 1 + 16 * widths.astype(np.int64))
        else:
            stream = np.full(target, it % 3, np.uint8)
        stream[pos] = widths
        damage = it % 3
        if damage == 1:
            stream[int(pos[int(rng.integers(0, widths.size))])] = int(rng.integers(33, 256))
        elif damage == 2 and target > 1:
            stream = stream[: int(rng.integers(1, target))]
        cap = widths.size + 3
        n, _, _, status = orc.stream_walk(stream, cap)
        bits, off, res = dev.stream_index(to_dev(stream), cap)
        res = res.tolist()
        assert res[2] == (1 if status else 0), it
        if status == 0:
            assert res[:2] == [n, stream.size - n], it
        p, good = 0, 0
        while p < stream.size and stream[p] <= 32 and p + 1 + 16 * int(stream[p]) <= stream.size:
            p += 1 + 16 * int(stream[p])
            good += 1
        good = min(good, cap)
        gb, go = bits.cpu().numpy()[:good], off.cpu().numpy()[:good]
        want = np.concatenate([[0], np.cumsum(1 + 16 * gb.astype(np.int64))])[:-1] if good else np.zeros(0, np.int64)
        assert np.array_equal(gb, stream[want]) and np.array_equal(go, want - np.arange(good)), it


def test_index_respects_max_blocks(dev, orc):
    """max_blocks smaller than the number of records: exactly max_blocks widths and offsets are written (guard zones
    on both sides of the two arrays stay untouched) and d_result still counts every record."""
    import torch

    from simdcomp_amd import binding as B

    L = B.lib()
    rng = np.random.default_rng(5)
    guard = 64
    for target in (5000, 16384, 70000, 400000, 1 << 20):
        widths, total = [], 0
        while True:
            w = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 20, 32]))
            if total + 1 + 16 * w > target:
                break
            widths.append(w)
            total += 1 + 16 * w
        widths = np.array(widths + [0] * (target - total), np.uint8)
        pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
        stream = rng.integers(0, 256, target, dtype=np.uint8)
        stream[pos] = widths
        n = widths.size
        for cap in (0, 1, n // 3, n - 1):
            bits = torch.full((cap + 2 * guard,), 0xAB, dtype=torch.uint8, device="cuda")
            off = torch.full((cap + 2 * guard,), -7, dtype=torch.int64, device="cuda")
            res = torch.zeros(3, dtype=torch.int64, device="cuda")
            ws = torch.empty(L.simdcomp_hip_stream_index_workspace(target), dtype=torch.uint8, device="cuda")
            B.check(L.simdcomp_hip_stream_index(B.ptr(to_dev(stream)), target, cap, bits[guard:].data_ptr(), off[guard:].data_ptr(),
                                                B.ptr(res), B.ptr(ws), torch.cuda.current_stream().cuda_stream), "stream_index")
            torch.cuda.synchronize()
            assert res.tolist() == [n, target - n, 0], (target, cap)
            assert (bits[:guard] == 0xAB).all() and (bits[guard + cap:] == 0xAB).all(), (target, cap)
            assert (off[:guard] == -7).all() and (off[guard + cap:] == -7).all(), (target, cap)
            assert np.array_equal(bits[guard:guard + cap].cpu().numpy(), widths[:cap]), (target, cap)
            assert np.array_equal(off[guard:guard + cap].cpu().numpy(), (pos - np.arange(n))[:cap]), (target, cap)


@pytest.mark.parametrize("pattern", ["all32", "all0", "alternate", "ramp", "one_wide_among_empty"])
@pytest.mark.parametrize("nblk", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_stream_copy_width_extremes(dev, pattern, nblk):
    """Arena <-> stream copies where a wave's 64 blocks hold anything from no payload vector to 2048 of them, with
    block counts around the wave and workgroup sizes; checked against the stream laid out in numpy, with guard bytes
    on both sides of both buffers."""
    import torch

    k = np.arange(nblk)
    widths = {"all32": np.full(nblk, 32), "all0": np.zeros(nblk, dtype=np.int64), "alternate": (k % 2) * 32, "ramp": k % 33,
              "one_wide_among_empty": np.where(k % 64 == 37, 32, 0)}[pattern].astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(16 * widths.astype(np.uint64))]).astype(np.uint64)
    nbytes = int(off[-1])
    rng = np.random.default_rng(nblk * 7 + len(pattern))
    arena = rng.integers(0, 256, nbytes + 16, dtype=np.uint8)
    want = np.empty(nbytes + nblk, dtype=np.uint8)
    for i in range(nblk):
        p = int(off[i]) + i
        want[p] = widths[i]
        want[p + 1:p + 1 + 16 * int(widths[i])] = arena[int(off[i]):int(off[i]) + 16 * int(widths[i])]
    d_bits, d_off, d_arena = to_dev(widths), to_dev(off[:nblk]), to_dev(arena)
    for shift in (0, 1, 7):
        buf = torch.full((want.size + 32,), 0xEE, dtype=torch.uint8, device="cuda")
        view = buf[8 + shift:8 + shift + want.size]
        dev.stream_build(d_arena, d_bits, d_off, nbytes, out=view)
        got = buf.cpu().numpy()
        assert np.array_equal(got[8 + shift:8 + shift + want.size], want)
        assert (got[:8 + shift] == 0xEE).all() and (got[8 + shift + want.size:] == 0xEE).all(), "wrote outside the stream"
        back = torch.full((nbytes + 48,), 0xCC, dtype=torch.uint8, device="cuda")
        dev.stream_unbuild(view, d_bits, d_off, nbytes, out=back[16:])
        got = back.cpu().numpy()
        assert np.array_equal(got[16:16 + nbytes], arena[:nbytes])
        assert (got[:16] == 0xCC).all() and (got[16 + nbytes:] == 0xCC).all(), "wrote outside the arena"


def test_one_call_entry_points_under_graph_capture(dev):
    """The batched entry points neither allocate nor synchronise, so a caller may capture them in a HIP graph:
    compress and decode of the byte stream captured once, replayed on new input in the same buffers."""
    import torch
    from simdcomp_amd import binding

    L, P = binding.lib(), binding.ptr
    nblk = 5000
    x = G.stream_corpus(321, nblk)
    d_x = to_dev(x)
    ws_k = torch.empty(L.simdcomp_hip_stream_compress_workspace(nblk), dtype=torch.uint8, device="cuda")
    nb = torch.zeros(1, dtype=torch.int64, device="cuda")
    strm = torch.zeros(513 * nblk, dtype=torch.uint8, device="cuda")
    ws_d = torch.empty(L.simdcomp_hip_stream_decode_workspace(513 * nblk, nblk), dtype=torch.uint8, device="cuda")
    out = torch.zeros(nblk * 128, dtype=torch.int32, device="cuda")
    r3 = torch.zeros(3, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # (warm-up outside the capture: module loading is not capturable)
        s = torch.cuda.current_stream().cuda_stream
        binding.check(L.simdcomp_hip_stream_compress(D1, 0, P(d_x), nblk, P(strm), P(nb), P(ws_k), s), "compress")
        binding.check(L.simdcomp_hip_stream_decode(D1, 0, P(strm), 513 * nblk, nblk, P(out), P(r3), P(ws_d), s), "decode")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        binding.check(L.simdcomp_hip_stream_compress(D1, 0, P(d_x), nblk, P(strm), P(nb), P(ws_k), s), "compress")
        # the decoder is given the whole buffer: what lies behind the stream has to parse as records too, so the
        # buffer is zero-filled (zero-width records) and max_blocks = nblk stops the index
        binding.check(L.simdcomp_hip_stream_decode(D1, 0, P(strm), 513 * nblk, nblk, P(out), P(r3), P(ws_d), s), "decode")
    for seed in (322, 323):
        y = G.stream_corpus(seed, nblk)
        d_x.copy_(to_dev(y))
        strm.zero_()
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(u32(out), y), seed
