"""Parity at the full sizes of the benchmark configurations.

The CPU oracle cannot walk 2^28..1.6*10^10 integers in test time, so these
tests use properties that hold for any size and are checked on the device
(decode(encode(x)) == x or x & mask, exact payload sizes, position-sensitive
checksums with a mismatch count), and compare a sample of blocks -- the first
and last ones and a thousand random ones -- byte for byte with the oracle.
They are the cases that exercise 64-bit indexing: more than 2^32 integers,
payload offsets beyond 2^32 bytes, millions of workgroups per launch."""
import numpy as np
import pytest

from simdcomp_amd import binding

pytestmark = pytest.mark.gpu

PLAIN, D1, FOR = 0, 1, 2


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from simdcomp_amd import codec

    return codec


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def sample_ids(nblk, count, seed):
    import torch

    rng = np.random.default_rng(seed)
    edge = np.concatenate([np.arange(min(8, nblk)), np.arange(max(0, nblk - 8), nblk)])
    idx = np.unique(np.concatenate([edge, rng.integers(0, nblk, count)]))
    return torch.from_numpy(idx).cuda()


def mismatches(dev, a, b):
    return int(dev.checksum(a, other=b)[1].item())


def randint32(n, seed):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    lo = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device="cuda", generator=g)
    hi = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device="cuda", generator=g)
    return lo | (hi << 16)


def test_random_order_2p28_all_widths(dev, orc):
    """Random 32-bit integers, 2^28 values, simdpack / simdpackwithoutmask / simdunpack at every width."""
    import torch

    n = 1 << 28
    nblk = n // 128
    x = randint32(n, 1)
    idx = sample_ids(nblk, 256, 2)
    xs = u32(x.view(-1, 128)[idx]).reshape(-1)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    for bit in range(33):
        payload = dev.pack(x, bit, masked=True)
        assert payload.numel() == nblk * 16 * bit
        dev.unpack(payload, nblk, bit, out=out)
        want = x if bit == 32 else x & ((1 << bit) - 1)
        assert mismatches(dev, out, want) == 0, bit
        if bit:
            got = payload.view(-1, 16 * bit)[idx].cpu().numpy().reshape(-1)
            assert np.array_equal(got, orc.batch_pack("plain", xs, bit, masked=True)), bit
    for bit in (1, 7, 19, 31):  # unmasked: out-of-range bits smear exactly like the reference
        payload = dev.pack(x, bit, masked=False)
        got = payload.view(-1, 16 * bit)[idx].cpu().numpy().reshape(-1)
        assert np.array_equal(got, orc.batch_pack("plain", xs, bit, masked=False)), bit


def sorted_ids(n, bit, seed):
    import torch

    gaps = randint32(n, seed)
    if bit < 32:
        gaps &= (1 << bit) - 1
    return torch.cumsum(gaps, 0, dtype=torch.int32)


def test_posting_lists_2p30_d1(dev, orc):
    """Sorted posting lists, 2^30 docIDs, at EVERY width 1..32: chained simdpackwithoutmaskd1 emitting the per-block
    offsets, simdunpackd1 back (compared on the device), simdmaxbitsd1, and ~1000 sampled blocks per width compared
    byte for byte with the oracle (payload, widths, and the decode of the oracle's own payload)."""
    import torch

    n = 1 << 30
    nblk = n // 128
    gaps32 = randint32(n, 10)
    idx = sample_ids(nblk, 1024, 3)
    offsets = torch.empty(nblk, dtype=torch.int32, device="cuda")
    back = torch.empty(n, dtype=torch.int32, device="cuda")
    for bit in range(1, 33):
        ids = torch.cumsum(gaps32 if bit == 32 else gaps32 & ((1 << bit) - 1), 0, dtype=torch.int32)
        payload = dev.packd1(ids, bit, initvalue=0, masked=False, initvalues_out=offsets)
        # the chained offsets are the last value of the previous block
        assert torch.equal(offsets[1:], ids[127::128][:-1]) and int(offsets[0].item()) == 0, bit
        dev.unpackd1(offsets, payload, nblk, bit, out=back)
        assert mismatches(dev, back, ids) == 0, bit
        widths = dev.maxbitsd1(ids, initvalue=0)
        assert int(widths.max().item()) <= bit, bit
        xs, so = u32(ids.view(-1, 128)[idx]).reshape(-1), u32(offsets[idx])
        got = payload.view(-1, 16 * bit)[idx].cpu().numpy().reshape(-1)
        want = orc.batch_pack("d1", xs, bit, seeds=so, masked=False)
        assert np.array_equal(got, want), bit
        assert np.array_equal(widths[idx].cpu().numpy(), orc.batch_maxbitsd1(so, xs)), bit
        assert np.array_equal(u32(back.view(-1, 128)[idx]).reshape(-1), orc.batch_unpack("d1", want, idx.numel(), bit, seeds=so)), bit
        del ids, payload, widths


def test_uniform_width_2p30_frame_of_reference(dev, orc):
    """2^30 integers at one uniform width per run, simdpackFOR / simdunpackFOR with a per-block frame anywhere in 32
    bits (value - frame wraps like uint32 where the cluster straddles 2^32): round trip on the device and sampled
    blocks against the oracle."""
    import torch

    n = 1 << 30
    nblk = n // 128
    noise = randint32(n, 30)
    bases = randint32(nblk, 31)
    idx = sample_ids(nblk, 512, 6)
    back = torch.empty(n, dtype=torch.int32, device="cuda")
    for bit in (1, 5, 8, 11, 17, 24, 31, 32):
        x = noise if bit == 32 else (noise & ((1 << bit) - 1)) + bases.repeat_interleave(128)
        lo = bases
        payload = dev.packFOR(lo, x, bit)
        assert payload.numel() == nblk * 16 * bit
        dev.unpackFOR(lo, payload, nblk, bit, out=back)
        assert mismatches(dev, back, x) == 0, bit
        xs, ss = u32(x.view(-1, 128)[idx]).reshape(-1), u32(lo[idx])
        got = payload.view(-1, 16 * bit)[idx].cpu().numpy().reshape(-1)
        want = orc.batch_pack("for", xs, bit, seeds=ss, masked=False)
        assert np.array_equal(got, want), bit
        assert np.array_equal(u32(back.view(-1, 128)[idx]).reshape(-1), orc.batch_unpack("for", want, idx.numel(), bit, seeds=ss)), bit
        del x, payload


def compare_mixed_sample(orc, mode, x, bits, off, payload, seeds, idx):
    xs = u32(x.view(-1, 128)[idx])
    bs = bits[idx].cpu().numpy()
    os_ = off[idx]
    ss = u32(seeds[idx])
    import torch

    for w in np.unique(bs):
        if w == 0:
            continue
        sel = np.nonzero(bs == w)[0]
        sel_t = torch.from_numpy(sel).cuda()
        byte_idx = os_[sel_t][:, None] + torch.arange(16 * int(w), device="cuda")[None, :]
        got = payload[byte_idx].cpu().numpy().reshape(-1)
        want = orc.batch_pack(mode, xs[sel].reshape(-1), int(w), seeds=ss[sel], masked=False)
        assert np.array_equal(got, want), (mode, int(w))


def test_clustered_2p30_frame_of_reference(dev, orc):
    """Clustered integers, 2^30 values: frame = simdmin, width = bits(max - min) per block
    (tests/unit.c:659-666), simdpackFOR / simdunpackFOR with per-block widths."""
    import torch

    n = 1 << 30
    nblk = n // 128
    wid = torch.minimum(torch.randint(0, 25, (nblk,), device="cuda"), torch.randint(0, 25, (nblk,), device="cuda"))
    x = randint32(n, 20)
    x &= ((1 << wid) - 1).to(torch.int32).repeat_interleave(128)
    x += randint32(nblk, 21).repeat_interleave(128)  # cluster bases anywhere in 32 bits; sums wrap like uint32
    lo, hi, bits = dev.maxmin(x)
    span = (hi.to(torch.int64) & 0xFFFFFFFF) - (lo.to(torch.int64) & 0xFFFFFFFF)
    want_bits = torch.where(span > 0, torch.floor(torch.log2(span.clamp(min=1).double())).long() + 1, torch.zeros_like(span))
    assert torch.equal(bits.long(), want_bits)
    off, total = dev.payload_offsets(bits)
    assert int(total.item()) == int(bits.long().sum().item()) * 16
    payload = torch.empty(int(total.item()) + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(FOR, x, bits, off, payload, initvalues=lo)
    back = dev.unpack_mixed(FOR, payload, bits, off, nblk, initvalues=lo)
    assert mismatches(dev, back, x) == 0
    compare_mixed_sample(orc, "for", x, bits, off, payload, lo, sample_ids(nblk, 1024, 4))


def test_mixed_width_d1_corpus_16e9(dev, orc):
    """1.6 * 10^10 docIDs (more than 2^32 integers, payload beyond 2^32 bytes), per-block widths 1..20 skewed small:
    simdmaxbitsd1 -> offsets scan -> pack -> unpack, the example.c:56-110 pipeline with stored offsets."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    n = 16_000_000_000 // 128 * 128
    if free < 190 * (1 << 30):
        pytest.skip("needs ~180 GiB of free HBM")
    nblk = n // 128
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    piece = 1 << 30
    carry = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i, start in enumerate(range(0, n, piece)):
        cnt = min(piece, n - start)
        w = torch.minimum(torch.randint(1, 21, (cnt // 128,), device="cuda"), torch.randint(1, 21, (cnt // 128,), device="cuda"))
        gaps = randint32(cnt, 100 + i)
        gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
        torch.cumsum(gaps, 0, dtype=torch.int32, out=ids[start:start + cnt])
        ids[start:start + cnt] += carry
        carry = ids[start + cnt - 1:start + cnt].clone()
        del gaps, w
    seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    assert 1 <= int(bits.max().item()) <= 20
    off, total = dev.payload_offsets(bits)
    nbytes = int(total.item())
    assert nbytes == int(bits.long().sum().item()) * 16 and nbytes > (1 << 32)
    assert int(off[-1].item()) == nbytes - 16 * int(bits[-1].item())
    payload = torch.empty(nbytes + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(D1, ids, bits, off, payload, initvalues=seeds)
    out = dev.unpack_mixed(D1, payload, bits, off, nblk, initvalues=seeds)
    assert mismatches(dev, out, ids) == 0
    compare_mixed_sample(orc, "d1", ids, bits, off, payload, seeds, sample_ids(nblk, 1024, 5))
    # the block-range shards of 8 GPUs decode to the same integers
    first, count = dev.partition(nblk, 8, 5)
    shard = dev.unpack_mixed(D1, payload, bits[first:first + count], off[first:first + count], count,
                             initvalues=seeds[first:first + count])
    assert mismatches(dev, shard, ids[first * 128:(first + count) * 128]) == 0
