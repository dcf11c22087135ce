"""Randomised cases against the oracle for the paths whose kernels depend on
the block count (workgroup tiling, ragged ends, wide layouts, generic-loop
hand-over between blocks, uniform seeds): shapes, widths, modes and data kinds
are drawn from a seeded generator.  SIMDCOMP_FUZZ_ROUNDS raises the number of
cases per seed (default 6)."""
import os

import numpy as np
import pytest

from simdcomp_amd import synth

pytestmark = pytest.mark.gpu
ROUNDS = int(os.environ.get("SIMDCOMP_FUZZ_ROUNDS", "6"))
COUNTS = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 1000, 2049, 4099]


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
        a = a.view(np.int32)
    elif a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a).cuda()


def to_np(t):
    a = t.cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else (a.view(np.uint64) if a.dtype == np.int64 else a)


def draw(rng, n, bit, wild):
    x = synth.rand_u32(int(rng.integers(1, 1 << 30)), n)
    return x if (wild or bit >= 32) else (x & synth.mask(bit))


@pytest.mark.parametrize("seed", range(4))
def test_wide_layouts(dev, orc, seed):
    import torch

    rng = np.random.default_rng(4000 + seed)
    for _ in range(ROUNDS):
        lanes = int(rng.choice([8, 16]))
        n = 32 * lanes
        nblk = int(rng.choice(COUNTS))
        bit = int(rng.integers(0, 33))
        masked, wild = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        x = draw(rng, nblk * n, bit, wild)
        want = np.concatenate([orc.lanes_pack(lanes, x[k * n:(k + 1) * n], bit, masked) for k in range(nblk)] + [np.zeros(0, np.uint8)])
        buf = torch.full((want.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        dev.pack_lanes(lanes, to_dev(x), bit, masked=masked, out=buf)
        got = to_np(buf)
        assert np.array_equal(got[:want.size], want) and (got[want.size:] == 0xA5).all(), (lanes, nblk, bit, masked, wild)
        out = torch.full((nblk * n + 16,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        dev.unpack_lanes(lanes, buf, nblk, bit, out=out)
        dec = to_np(out)
        stride = 4 * lanes * bit
        ref = np.concatenate([orc.lanes_unpack(lanes, want[k * stride:(k + 1) * stride], bit) for k in range(nblk)])
        assert np.array_equal(dec[:nblk * n], ref) and (dec[nblk * n:] == 0x5A5A5A5A).all(), (lanes, nblk, bit)
        assert np.array_equal(to_np(dev.maxbits_lanes(lanes, to_dev(x))),
                              np.array([orc.lanes_maxbits(lanes, x[k * n:(k + 1) * n]) for k in range(nblk)], np.uint8))


@pytest.mark.parametrize("seed", range(4))
def test_reductions(dev, orc, seed):
    rng = np.random.default_rng(5000 + seed)
    for _ in range(ROUNDS):
        nblk = int(rng.choice(COUNTS))
        bit = int(rng.integers(0, 33))
        x = draw(rng, nblk * 128, bit, bool(rng.integers(0, 2)))
        if rng.integers(0, 2):
            x = (np.cumsum(x.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        xd = to_dev(x)
        assert np.array_equal(to_np(dev.maxbits(xd)), orc.batch_maxbits(x))
        seeds = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk)
        assert np.array_equal(to_np(dev.maxbitsd1(xd, initvalues=to_dev(seeds))), orc.batch_maxbitsd1(seeds, x))
        import torch

        chained = synth.block_offsets(x, 77)
        seeds_out = torch.full((nblk + 4,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        got = dev.maxbitsd1(xd, initvalue=77, initvalues_out=seeds_out)
        assert np.array_equal(to_np(got), orc.batch_maxbitsd1(chained, x)), nblk
        so = to_np(seeds_out)
        assert np.array_equal(so[:nblk], chained) and (so[nblk:] == 0x5A5A5A5A).all()
        lo, hi, bits = dev.maxmin(xd)
        wlo, whi = orc.batch_minmax(x)
        assert np.array_equal(to_np(lo), wlo) and np.array_equal(to_np(hi), whi)
        assert np.array_equal(to_np(bits), np.array([orc.bits(int(h) - int(l)) for l, h in zip(wlo, whi)], np.uint8))


@pytest.mark.parametrize("seed", range(4))
def test_frame_of_reference_lengths(dev, orc, seed):
    """simdpackFOR_length / simdunpackFOR_length: full blocks on the fast kernels, ragged end and partial last block on
    the flexible one, joined by the hand-over between blocks (out-of-range data makes it visible)."""
    import torch

    rng = np.random.default_rng(6000 + seed)
    for _ in range(ROUNDS):
        n = int(rng.choice([1, 127, 128, 129, 1023, 1024, 1025, 4096, 4100, 8191, 33000, 70001]))
        bit = int(rng.integers(0, 33))
        base = int(rng.integers(0, 1 << 32))
        x = draw(rng, n, bit, bool(rng.integers(0, 2)))
        if rng.integers(0, 2):
            x = (x + np.uint32(base)).astype(np.uint32)
        want = orc.simdpackFOR_length(base, x, bit)
        buf = torch.full((want.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        dev.packFOR_length(base, to_dev(x), bit, out=buf)
        got = to_np(buf)
        assert np.array_equal(got[:want.size], want) and (got[want.size:] == 0xA5).all(), (n, bit)
        out = torch.full((n + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        dev.unpackFOR_length(base, buf, n, bit, out=out)
        dec = to_np(out)
        wdec, _ = orc.simdunpackFOR_length(base, want, n, bit)
        assert np.array_equal(dec[:n], wdec) and (dec[n:] == 0x5A5A5A5A).all(), (n, bit)


@pytest.mark.parametrize("seed", range(4))
def test_per_block_widths_and_stream(dev, orc, seed):
    import torch

    rng = np.random.default_rng(7000 + seed)
    for _ in range(ROUNDS):
        nblk = int(rng.choice(COUNTS))
        top = int(rng.choice([4, 12, 20, 32]))
        x = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk * 128)
        widths = rng.integers(0, top + 1, nblk)
        for k in range(nblk):
            if widths[k] < 32:
                x[k * 128:(k + 1) * 128] &= synth.mask(int(widths[k]))
        ids = (np.cumsum(x.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        want = orc.stream_compress(ids, d1=True, initvalue=5)
        buf, nbytes = dev.stream_compress(1, to_dev(ids), initvalue=5)
        T = int(nbytes.item())
        assert T == want.size and np.array_equal(to_np(buf)[:T], want), (nblk, top)
        out, res = dev.stream_decode(1, buf[:T].clone(), nblk, initvalue=5)
        assert res.tolist() == [nblk, T - nblk, 0]
        assert np.array_equal(to_np(out), ids), (nblk, top)
        # plain coding over the same integers
        want = orc.stream_compress(x, d1=False)
        buf, nbytes = dev.stream_compress(0, to_dev(x))
        T = int(nbytes.item())
        assert T == want.size and np.array_equal(to_np(buf)[:T], want)
        out, res = dev.stream_decode(0, buf[:T].clone(), nblk)
        assert res.tolist() == [nblk, T - nblk, 0] and np.array_equal(to_np(out), x)
