"""The scalar restatement (oracle/) against the committed golden fixtures, which
were produced by the compiled reference.  CPU only; runs anywhere."""
import numpy as np
import pytest

import golden_util as G


def fnv_py(data):
    h = 0xCBF29CE484222325
    for byte in bytes(data):
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def test_fnv_helper(orc):
    a = np.arange(1000, dtype=np.uint32)
    assert orc.fnv1a64(a) == fnv_py(a.tobytes())
    assert orc.fnv1a64(np.zeros(0, np.uint8)) == 0xCBF29CE484222325


def _words(p, n=8):
    return p.view(np.uint32)[:n].tolist()


def test_known_answers(orc):
    """Known answers recorded from the compiled reference when it was surveyed:
    the leading compressed words of a few closed-form inputs."""
    x = np.arange(128, dtype=np.uint32)
    b = orc.maxbits(x)
    assert b == 7
    assert _words(orc.simdpackwithoutmask(x, b)) == [0x01820200, 0x11A24281, 0x21C28302, 0x31E2C383,
                                                     0x203860A1, 0xA13A64A9, 0x223C68B1, 0xA33E6CB9]

    x = np.arange(1, 129, dtype=np.uint32)
    assert orc.simdmaxbitsd1(0, x) == 1
    assert _words(orc.simdpackwithoutmaskd1(0, x, 1), 4) == [0xFFFFFFFF] * 4

    i = np.arange(128, dtype=np.uint64)
    x = (3 * i * i + 5).astype(np.uint32)
    b = orc.simdmaxbitsd1(2, x)
    assert b == 10
    p = orc.simdpackwithoutmaskd1(2, x, b)
    assert _words(p) == [0x42D05403, 0xC3306C03, 0x43908409, 0xC3F09C0F,
                         0xD1D45D11, 0x31EC6312, 0x92046914, 0xF21C6F15]
    assert np.array_equal(orc.simdunpackd1(2, p, b), x)

    x = (1000 + 3 * np.arange(128)).astype(np.uint32)
    lo, hi = orc.simdmaxmin(x)
    assert (lo, hi) == (1000, 1381)
    b = orc.bits(hi - lo)
    assert b == 9
    assert _words(orc.simdpackFOR(lo, x, b)) == [0x20601800, 0x386C1E03, 0x50782406, 0x68842A09,
                                                 0x12078301, 0x92C7E331, 0x13884361, 0x9448A391]

    x = ((np.arange(1, 129, dtype=np.uint64) * 0x9E3779B9) & 0xFFFFFFFF).astype(np.uint32)
    assert _words(orc.simdpack(x, 3)) == [0x69A69A69, 0xB2CB2CB2, 0xFBEFBEFB, 0x04104104,
                                          0x9A69A69A, 0x2CB2CB2C, 0xBEFBEFBE, 0x41041041]
    assert _words(orc.simdpack(x, 11)) == [0xE044E9B9, 0x4E92B372, 0xBCE07D2B, 0x2B2E46E4,
                                           0x16D49CCB, 0xF370203C, 0xCF8BB3AE, 0xAC274721]
    assert _words(orc.simdpack(x, 25)) == [0x3A3779B9, 0xAC6EF372, 0x1EA66D2B, 0x90DDE6E4,
                                           0x1E062AC1, 0x04EA99B4, 0xEBCF08A8, 0xD2B3779B]
    assert _words(orc.simdpack(x, 32), 4) == [0x9E3779B9, 0x3C6EF372, 0xDAA66D2B, 0x78DDE6E4]

    assert orc.compressedbytes(9999, 14) == 17504
    assert orc.compressedbytes(130, 7) == 128
    assert orc.compressedbytes(5, 32) == 20


@pytest.mark.parametrize("bit", range(33))
def test_blocks(orc, bit):
    g = G.load("blocks.npz")
    for kind in range(3):
        x = g["x"][bit, kind].copy()
        iv = int(g["init"][bit, kind])
        outs = [orc.simdpack(x, bit), orc.simdpackwithoutmask(x, bit), orc.simdpackd1(iv, x, bit),
                orc.simdpackwithoutmaskd1(iv, x, bit), orc.simdpackFOR(iv, x, bit)]
        for j, o in enumerate(outs):
            want = g["packed"][bit, kind, j, :16 * bit]
            assert np.array_equal(o, want), (G.PACK_NAMES[j], bit, kind)
            assert np.array_equal(orc.simdunpack(want.copy(), bit), g["unpacked"][bit, kind, j, 0])
            assert np.array_equal(orc.simdunpackd1(iv, want.copy(), bit), g["unpacked"][bit, kind, j, 1])
            assert np.array_equal(orc.simdunpackFOR(iv, want.copy(), bit), g["unpacked"][bit, kind, j, 2])
        lo, hi = orc.simdmaxmin(x)
        assert [orc.maxbits(x), orc.simdmaxbitsd1(iv, x), lo, hi] == g["red"][bit, kind].tolist()


def test_lengths(orc):
    g = G.load("lengths.npz")
    base = int(g["base"])
    for n, bit, kind, nbytes, hp, hpf, hu, huf in g["table"].tolist():
        x = G.length_input(n, bit, kind)
        p = orc.simdpack_length(x, bit)
        pf = orc.simdpackFOR_length(base, x, bit)
        assert len(p) == nbytes == orc.compressedbytes(n, bit)
        assert orc.fnv1a64(p) == hp and orc.fnv1a64(pf) == hpf, (n, bit, kind)
        u, used = orc.simdunpack_length(p, n, bit)
        uf, usedf = orc.simdunpackFOR_length(base, pf, n, bit)
        assert used == nbytes and usedf == nbytes
        assert orc.fnv1a64(u) == hu and orc.fnv1a64(uf) == huf, (n, bit, kind)
        if n in G.FULL_LENGTHS:
            key = f"n{n}_b{bit}_k{kind}"
            assert np.array_equal(p, g[key + "_p"]) and np.array_equal(pf, g[key + "_pf"])
            assert np.array_equal(u, g[key + "_u"]) and np.array_equal(uf, g[key + "_uf"])


def test_corpus_digests(orc):
    g = G.load("corpus.npz")
    nblk = int(g["nblocks"])
    for mode, bit, masked, hpay, hdec in g["table"].tolist():
        if mode == 0:
            v = G.corpus_plain(bit, masked, nblk)
            pay = orc.batch_pack("plain", v, bit, masked=bool(masked))
            dec = orc.batch_unpack("plain", pay, nblk, bit)
        elif mode == 1:
            v, offs = G.corpus_d1(bit, nblk)
            pay = orc.batch_pack("d1", v, bit, seeds=offs, masked=bool(masked))
            dec = orc.batch_unpack("d1", pay, nblk, bit, seeds=offs)
        else:
            v, base = G.corpus_for(bit, nblk)
            pay = orc.batch_pack("for", v, bit, seeds=base)
            dec = orc.batch_unpack("for", pay, nblk, bit, seeds=base)
        assert orc.fnv1a64(pay) == hpay, (mode, bit, masked)
        assert orc.fnv1a64(dec) == hdec, (mode, bit, masked)
    v, offs = G.corpus_mixed(nblk)
    bits = orc.batch_maxbitsd1(offs, v)
    pos = np.concatenate([[0], np.cumsum(bits.astype(np.uint64) * 16)]).astype(np.uint64)
    pay = orc.batch_pack("d1", v, seeds=offs, bits=bits, off=pos[:-1].copy(), out_bytes=int(pos[-1]))
    dec = orc.batch_unpack("d1", pay, nblk, seeds=offs, bits=bits, off=pos[:-1].copy())
    assert [orc.fnv1a64(bits), orc.fnv1a64(pay), orc.fnv1a64(dec)] == g["mixed"].tolist()
    assert np.array_equal(dec, v)
