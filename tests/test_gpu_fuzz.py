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


@pytest.mark.parametrize("seed", range(4))
def test_query_batches(dev, orc, seed):
    """Batched search / select with the batch's structure drawn at random: how many queries, how long the runs of
    queries on one block are, ascending or shuffled block order, d1 or frame of reference, one width or per-block
    widths (0-bit and raw 32-bit blocks included), lengths 0..128, the offset vector simdsearchd1 leaves behind."""
    import torch

    rng = np.random.default_rng(8000 + seed)
    for _ in range(ROUNDS):
        nblk = int(rng.choice([1, 2, 3, 17, 64, 65, 300]))
        mode = int(rng.choice([1, 2]))  # d1, frame of reference
        uniform = bool(rng.integers(0, 2))
        widths = np.full(nblk, int(rng.integers(0, 33)), np.int64) if uniform else rng.integers(0, 33, nblk)
        x = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk * 128).reshape(nblk, 128)
        for k in range(nblk):
            if widths[k] < 32:
                x[k] &= synth.mask(int(widths[k]))
        seeds = synth.rand_u32(int(rng.integers(1, 1 << 30)), nblk) >> np.uint32(rng.integers(0, 8))
        if mode == 1:
            vals = np.empty_like(x)
            for k in range(nblk):  # gaps -> values, from the block's own seed (a raw block keeps its words)
                vals[k] = x[k] if widths[k] == 32 else (np.cumsum(x[k].astype(np.uint64)) + np.uint64(seeds[k])).astype(np.uint32)
        else:
            vals = (x + seeds[:, None]).astype(np.uint32)
            raw = widths == 32
            vals[raw] = x[raw]
        bits = to_dev(widths.astype(np.uint8))
        off, total = dev.payload_offsets(bits)
        arena = torch.zeros(int(total.item()) + 16, dtype=torch.uint8, device="cuda")
        dev.pack_mixed(mode, to_dev(vals.reshape(-1)), bits, off, arena, initvalues=to_dev(seeds))
        h_arena, h_off = to_np(arena), to_np(off)
        nq = int(rng.choice([1, 5, 63, 64, 65, 200, 1000]))
        runs = rng.integers(1, int(rng.choice([2, 5, 70])) + 1, nq)  # run lengths
        qb = np.repeat(rng.integers(0, nblk, nq), runs)[:nq]
        if rng.integers(0, 2):
            qb = np.sort(qb)
        qb = qb.astype(np.uint64)
        lengths = rng.integers(0, 129, nq).astype(np.int32)
        if mode == 1:
            lengths[widths[qb.astype(np.int64)] == 32] = np.maximum(lengths[widths[qb.astype(np.int64)] == 32], 1)
        keys = (vals[qb.astype(np.int64), rng.integers(0, 128, nq)] + rng.integers(0, 3, nq).astype(np.uint32) - np.uint32(1)).astype(np.uint32)
        slots = rng.integers(0, 128, nq).astype(np.int32)
        kw = dict(initvalues=to_dev(seeds), query_blocks=to_dev(qb))
        if uniform:
            kw["bit"] = int(widths[0])
        else:
            kw.update(bits=bits, offsets=off)
        got = to_np(dev.select(mode, arena, to_dev(slots), **kw))
        want_last4 = mode == 1 and bool(rng.integers(0, 2))
        res = dev.search(mode, arena, to_dev(keys), lengths=to_dev(lengths), want_last4=want_last4, **kw)
        pos, val = to_np(res[0]).view(np.int32), to_np(res[1])
        for q in range(nq):
            k, b = int(qb[q]), int(widths[qb[q]])
            comp = np.concatenate([h_arena[int(h_off[k]):int(h_off[k]) + 16 * b], np.zeros(16, np.uint8)])
            if mode == 1:
                assert int(got[q]) == orc.simdselectd1(int(seeds[k]), comp, b, int(slots[q])), (q, k, b)
                want = orc.simdsearchwithlengthd1(int(seeds[k]), comp, b, int(lengths[q]), int(keys[q]))
                assert (int(pos[q]), int(val[q])) == tuple(int(v) for v in want), (q, k, b, int(lengths[q]))
            else:
                assert int(got[q]) == orc.simdselectFOR(int(seeds[k]), comp, b, int(slots[q])), (q, k, b)
                wpos, wval = orc.simdsearchwithlengthFOR(int(seeds[k]), comp, b, int(lengths[q]), int(keys[q]))
                assert (int(pos[q]), int(val[q])) == (wpos, 0 if wpos == 128 else wval), (q, k, b, int(lengths[q]))
        if want_last4:
            # the offset vector is defined for searches over the whole block
            full = np.nonzero(lengths == 128)[0]
            last4 = to_np(res[2]).reshape(-1, 4)
            for q in full[:50]:
                k, b = int(qb[q]), int(widths[qb[q]])
                comp = np.concatenate([h_arena[int(h_off[k]):int(h_off[k]) + 16 * b], np.zeros(16, np.uint8)])
                o4 = np.array([0, 0, 0, seeds[k]], np.uint32)
                wpos, wval, wo4 = orc.simdsearchd1(o4, comp, b, int(keys[q]))
                if b != 0:  # (a 0-bit block leaves the caller's vector untouched: the batched form reports the seed four times)
                    assert np.array_equal(last4[q], wo4), (q, k, b)
