"""The caller loops of example.c (byte stream of [width byte][payload] records,
chained differential decode without stored offsets): the restatement's block
routines inside those loops against the compiled reference's, and both against
the committed streams the reference produced."""
import numpy as np
import pytest

import golden_util as G


def test_golden_streams_restatement(orc):
    g = G.load("stream.npz")
    nblk = int(g["nblocks"])
    x = G.stream_corpus(900, nblk)
    assert np.array_equal(orc.stream_compress(x, d1=True), g["d1"])
    assert np.array_equal(orc.stream_compress(x, d1=False), g["plain"])
    assert np.array_equal(orc.stream_decode(g["d1"], nblk, d1=True), x)
    assert np.array_equal(orc.stream_decode(g["plain"], nblk, d1=False), x)


def test_stream_loops_against_reference(orc, ref):
    for seed, nblk in ((1, 1), (2, 7), (3, 300)):
        x = G.stream_corpus(seed, nblk)
        for d1 in (True, False):
            for init in ((0, 77) if d1 else (0,)):
                a = orc.stream_compress(x, d1=d1, initvalue=init)
                b = orc.stream_compress(x, d1=d1, initvalue=init, codec=ref)
                assert np.array_equal(a, b)
                assert np.array_equal(orc.stream_decode(a, nblk, d1=d1, initvalue=init, codec=ref), x)
                assert np.array_equal(orc.stream_decode(a, nblk, d1=d1, initvalue=init), x)


def test_stream_walk(orc):
    g = G.load("stream.npz")
    nblk = int(g["nblocks"])
    x = G.stream_corpus(900, nblk)
    s = g["d1"]
    n, bits, off, status = orc.stream_walk(s, nblk)
    assert (n, status) == (nblk, 0)
    offs = np.concatenate([[0], x[127::128][:-1]]).astype(np.uint32)
    assert np.array_equal(bits, orc.batch_maxbitsd1(offs, x))
    want = np.concatenate([[0], np.cumsum(bits.astype(np.uint64) * 16)[:-1]]).astype(np.uint64)
    assert np.array_equal(off, want)
    assert s.size == nblk + int(bits.astype(np.uint64).sum()) * 16
    # a cut stream and a stream with an impossible width byte are reported
    assert orc.stream_walk(s[:-1], nblk)[3] == 1
    bad = s.copy()
    bad[0] = 33
    assert orc.stream_walk(bad, nblk)[3] == 1
    assert orc.stream_walk(s[:0], nblk)[::3] == (0, 0)
