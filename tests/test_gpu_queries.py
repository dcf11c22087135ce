"""Batched select / search / scan on the device (simdcomp_hip_select, _search,
_scand1) through the C ABI: against the answers the reference gave
(tests/golden/queries.npz) and against the oracle on seeded corpora, uniform and
per-block widths, arena and byte-stream layouts, sorted and unsorted data."""
import numpy as np
import pytest

import golden_util as G
from simdcomp_amd import synth

pytestmark = pytest.mark.gpu

PLAIN, D1, FOR = 0, 1, 2


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


def test_golden_queries(dev, orc):
    """Every (block, query) pair of the fixture as one batch per width and kind."""
    import torch

    g = G.load("queries.npz")
    for bit in range(33):
        for kind in range(2):
            x, init = G.query_block(bit, kind)
            d1 = orc.simdpackwithoutmaskd1(init, x, bit)
            fr = orc.simdpackFOR(init, x, bit)
            keys = G.query_keys(orc.simdunpackd1(init, d1, bit), bit, kind)
            pad = np.zeros(16, np.uint8)
            d_d1, d_fr = to_dev(np.concatenate([d1, pad])), to_dev(np.concatenate([fr, pad]))
            slots = to_dev(np.array(G.QUERY_SLOTS, np.int32))
            zero = torch.zeros(len(G.QUERY_SLOTS), dtype=torch.int64, device="cuda")
            got = dev.select(D1, d_d1, slots, bit=bit, initvalue=init, query_blocks=zero)
            assert np.array_equal(u32(got), g["select"][bit, kind, 0])
            got = dev.select(FOR, d_fr, slots & 127, bit=bit, initvalue=init, query_blocks=zero)
            assert np.array_equal(u32(got), g["select"][bit, kind, 1])
            kk = np.repeat(keys, len(G.QUERY_LENGTHS))
            ll = np.tile(np.array(G.QUERY_LENGTHS, np.int32), keys.size)
            zero = torch.zeros(kk.size, dtype=torch.int64, device="cuda")
            pos, val = dev.search(D1, d_d1, to_dev(kk), bit=bit, initvalue=init, query_blocks=zero, lengths=to_dev(ll))
            want = g["search"][bit, kind, 0].reshape(-1, 2)
            assert np.array_equal(pos.cpu().numpy(), want[:, 0]) and np.array_equal(u32(val), want[:, 1].astype(np.uint32)), bit
            pos, val = dev.search(FOR, d_fr, to_dev(kk), bit=bit, initvalue=init, query_blocks=zero,
                                  lengths=to_dev(np.minimum(ll, 127)))
            want = g["search"][bit, kind, 1].reshape(-1, 2)
            assert np.array_equal(pos.cpu().numpy(), want[:, 0]) and np.array_equal(u32(val), want[:, 1].astype(np.uint32)), bit
            zero = torch.zeros(keys.size, dtype=torch.int64, device="cuda")
            pos, val, last4 = dev.search(D1, d_d1, to_dev(keys), bit=bit, initvalue=init, query_blocks=zero, want_last4=True)
            want = g["searchd1"][bit, kind]
            assert np.array_equal(pos.cpu().numpy(), want[:, 0]) and np.array_equal(u32(val), want[:, 1].astype(np.uint32))
            assert np.array_equal(u32(last4).reshape(-1, 4), want[:, 2:].astype(np.uint32)), (bit, kind)
            got = dev.scand1(d_d1, 1, bit=bit, initvalue=init)
            assert np.array_equal(u32(got), g["scan"][bit, kind])


def mixed_corpus(dev, nblk, seed):
    """Per-block widths (0-bit and 32-bit blocks included), packed on the device in arena and stream layout."""
    import torch

    x = G.stream_corpus(seed, nblk)
    d_x = to_dev(x)
    seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
    bits = dev.maxbitsd1(d_x, initvalue=0, initvalues_out=seeds)
    off, total = dev.payload_offsets(bits)
    nbytes = int(total.item())
    arena = torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(D1, d_x, bits, off, arena, initvalues=seeds)
    stream = torch.zeros(nbytes + nblk + 3, dtype=torch.uint8, device="cuda")[3:]
    dev.pack_stream(D1, d_x, bits, off, stream, initvalues=seeds)
    return x, seeds, bits, off, arena, stream


@pytest.mark.parametrize("order", ["random", "sorted", "hot", "descending"])
def test_batched_queries_mixed_widths(dev, orc, order):
    """(block, key) and (block, slot) queries over a mixed-width d1 corpus, both layouts: a random block per query;
    the same batch sorted by block (neighbouring queries on one block share its decode: runs of every length,
    across the 64-query groups the kernel works in); a batch on a handful of blocks (runs far longer than a group);
    and sorted downwards (runs, but no ascending order to lean on)."""
    import torch

    nblk, nq = 3000, 20011
    x, seeds, bits, off, arena, stream = mixed_corpus(dev, nblk, 61)
    h_bits, h_off, h_seeds, h_arena = bits.cpu().numpy(), off.cpu().numpy(), u32(seeds), arena.cpu().numpy()
    qb = (synth.rand_u32(1, nq) % np.uint32(nblk)).astype(np.uint64)
    if order == "hot":
        qb = (qb % np.uint64(37)) * np.uint64(81)
    if order != "random":
        qb = np.sort(qb)
    if order == "descending":
        qb = qb[::-1].copy()
    slots = (synth.rand_u32(2, nq) % np.uint32(256)).astype(np.int32)
    lengths = (synth.rand_u32(3, nq) % np.uint32(129)).astype(np.int32)
    raw = h_bits[qb.astype(np.int64)] == 32
    lengths[raw] = np.maximum(lengths[raw], 1)  # the reference is undefined for length 0 on a raw block
    # keys near the values of the addressed block so that hits, misses and ties all occur
    pick = x.reshape(nblk, 128)[qb.astype(np.int64), synth.rand_u32(4, nq) % np.uint32(128)]
    keys = (pick + (synth.rand_u32(5, nq) % np.uint32(3)) - np.uint32(1)).astype(np.uint32)
    want_sel = np.zeros(nq, np.uint32)
    want_pos = np.zeros(nq, np.int32)
    want_val = np.zeros(nq, np.uint32)
    for q in range(nq):
        k = int(qb[q])
        comp = h_arena[int(h_off[k]):int(h_off[k]) + 16 * int(h_bits[k])]
        want_sel[q] = orc.simdselectd1(int(h_seeds[k]), comp, int(h_bits[k]), int(slots[q]))
        want_pos[q], want_val[q] = orc.simdsearchwithlengthd1(int(h_seeds[k]), comp, int(h_bits[k]), int(lengths[q]), int(keys[q]))
    for payload, layout in ((arena, False), (stream, True)):
        got = dev.select(D1, payload, to_dev(slots), bits=bits, offsets=off, initvalues=seeds, query_blocks=to_dev(qb),
                         stream_layout=layout)
        assert np.array_equal(u32(got), want_sel)
        pos, val = dev.search(D1, payload, to_dev(keys), bits=bits, offsets=off, initvalues=seeds, query_blocks=to_dev(qb),
                              lengths=to_dev(lengths), stream_layout=layout)
        bad = np.nonzero((pos.cpu().numpy() != want_pos) | (u32(val) != want_val))[0]
        assert bad.size == 0, [(int(q), int(h_bits[int(qb[q])]), int(lengths[q]), int(keys[q]), int(pos[q]), int(want_pos[q]),
                                int(u32(val)[q]), int(want_val[q])) for q in bad[:5]]
        last4 = dev.scand1(payload, nblk, bits=bits, offsets=off, initvalues=seeds, stream_layout=layout)
        assert np.array_equal(u32(last4).reshape(nblk, 4), x.reshape(nblk, 128)[:, 124:])
        # the same queries with the blocks' metadata taken from a block directory
        directory = dev.block_directory(bits, off, initvalues=seeds, stream_layout=layout)
        got = dev.select_directory(D1, payload, directory, to_dev(slots), query_blocks=to_dev(qb))
        assert np.array_equal(u32(got), want_sel)
        dpos, dval = dev.search_directory(D1, payload, directory, to_dev(keys), query_blocks=to_dev(qb), lengths=to_dev(lengths))
        assert np.array_equal(dpos.cpu().numpy(), want_pos) and np.array_equal(u32(dval), want_val)
        _, _, l4 = dev.search(D1, payload, to_dev(keys), bits=bits, offsets=off, initvalues=seeds, query_blocks=to_dev(qb),
                              stream_layout=layout, want_last4=True)
        _, _, dl4 = dev.search_directory(D1, payload, directory, to_dev(keys), query_blocks=to_dev(qb), want_last4=True)
        assert torch.equal(l4, dl4)


@pytest.mark.parametrize("bit", [0, 1, 5, 13, 31, 32])
def test_batched_queries_uniform_for(dev, orc, bit):
    """Frame-of-reference blocks at one width: one query per block (no block list), sorted inside each block."""
    nblk = 1009
    x, base = synth.clustered(300 + bit, nblk, bit)
    x = np.sort(x.reshape(nblk, 128), axis=1).reshape(-1)
    payload = dev.packFOR(to_dev(base), to_dev(x), bit)
    slots = (synth.rand_u32(7, nblk) % np.uint32(128)).astype(np.int32)
    got = dev.select(FOR, payload, to_dev(slots), bit=bit, initvalues=to_dev(base))
    assert np.array_equal(u32(got), x.reshape(nblk, 128)[np.arange(nblk), slots])
    keys = (x.reshape(nblk, 128)[np.arange(nblk), synth.rand_u32(8, nblk) % np.uint32(128)]
            + (synth.rand_u32(9, nblk) % np.uint32(3)) - np.uint32(1)).astype(np.uint32)
    lengths = (1 + synth.rand_u32(10, nblk) % np.uint32(127)).astype(np.int32)
    pos, val = dev.search(FOR, payload, to_dev(keys), bit=bit, initvalues=to_dev(base), lengths=to_dev(lengths))
    h = payload.cpu().numpy()
    for k in range(nblk):
        want = orc.simdsearchwithlengthFOR(int(base[k]), h[k * 16 * bit:(k + 1) * 16 * bit], bit, int(lengths[k]), int(keys[k]))
        assert (int(pos[k]), int(u32(val[k:k + 1])[0])) == want, (bit, k)


@pytest.mark.parametrize("mode", [FOR, PLAIN])
def test_batched_queries_for_and_plain_with_runs(dev, orc, mode):
    """Frame-of-reference and plain blocks with per-block widths, batches in which neighbouring queries share a block
    (runs of every length) and batches that do not: simdselectFOR / simdsearchwithlengthFOR per query on the host
    as the reference would be called (plain = frame base 0), blocks sorted and unsorted inside."""
    import torch

    nblk, nq = 700, 9001
    widths = synth.skewed_widths(91, nblk, 0, 32)
    x, base = synth.clustered(92, nblk, widths)
    if mode == PLAIN:
        x = (x.reshape(nblk, 128) - base[:, None]).astype(np.uint32).reshape(-1)
        base = np.zeros(nblk, np.uint32)
    xs = x.reshape(nblk, 128).copy()
    xs[::2] = np.sort(xs[::2], axis=1)  # every other block sorted: the bisection is defined on any content
    x = xs.reshape(-1)
    bits = to_dev(widths)
    off, total = dev.payload_offsets(bits)
    arena = torch.zeros(int(total.item()) + 16, dtype=torch.uint8, device="cuda")
    dev.pack_mixed(mode, to_dev(x), bits, off, arena, initvalues=to_dev(base) if mode == FOR else None)
    h_arena, h_off = arena.cpu().numpy(), off.cpu().numpy()
    for order in ("random", "sorted", "hot"):
        qb = (synth.rand_u32(93, nq) % np.uint32(nblk)).astype(np.uint64)
        if order == "hot":
            qb = (qb % np.uint64(11)) * np.uint64(60)
        if order != "random":
            qb = np.sort(qb)
        slots = (synth.rand_u32(94, nq) % np.uint32(128)).astype(np.int32)
        lengths = (synth.rand_u32(95, nq) % np.uint32(129)).astype(np.int32)
        pick = xs[qb.astype(np.int64), synth.rand_u32(96, nq) % np.uint32(128)]
        keys = (pick + (synth.rand_u32(97, nq) % np.uint32(3)) - np.uint32(1)).astype(np.uint32)
        kw = dict(bits=bits, offsets=off, query_blocks=to_dev(qb))
        if mode == FOR:
            kw["initvalues"] = to_dev(base)
        got = dev.select(mode, arena, to_dev(slots), **kw)
        assert np.array_equal(u32(got), xs[qb.astype(np.int64), slots]), (mode, order)
        pos, val = dev.search(mode, arena, to_dev(keys), lengths=to_dev(lengths), **kw)
        directory = dev.block_directory(bits, off, initvalues=kw.get("initvalues"))
        assert torch.equal(got, dev.select_directory(mode, arena, directory, to_dev(slots), query_blocks=to_dev(qb)))
        dpos, dval = dev.search_directory(mode, arena, directory, to_dev(keys), query_blocks=to_dev(qb), lengths=to_dev(lengths))
        assert torch.equal(pos, dpos) and torch.equal(val, dval)
        pos, val = pos.cpu().numpy(), u32(val)
        for q in range(0, nq, 7):
            k = int(qb[q])
            b = int(widths[k])
            comp = np.concatenate([h_arena[int(h_off[k]):int(h_off[k]) + 16 * b], np.zeros(16, np.uint8)])
            wpos, wval = orc.simdsearchwithlengthFOR(int(base[k]), comp, b, int(lengths[q]), int(keys[q]))
            if wpos == 128:
                wval = 0  # the reference reads past the block there; the batched form reports 0
            assert (int(pos[q]), int(val[q])) == (wpos, wval), (mode, order, q, b)
