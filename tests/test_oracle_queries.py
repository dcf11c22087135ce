"""simdselectd1 / simdsearchwithlengthd1 / simdsearchd1 / simdscand1 and the FOR
select / search: the restatement against the answers the reference gave
(tests/golden/queries.npz) and, where the compiled reference is present,
against the reference itself on more inputs."""
import numpy as np
import pytest

import golden_util as G


def cases():
    for bit in range(33):
        for kind in range(2):
            yield bit, kind


def packed(orc, bit, kind):
    x, init = G.query_block(bit, kind)
    d1 = orc.simdpackwithoutmaskd1(init, x, bit)
    fr = orc.simdpackFOR(init, x, bit)
    dec = orc.simdunpackd1(init, d1, bit)
    return x, init, d1, fr, G.query_keys(dec, bit, kind)


def test_golden_queries(orc):
    g = G.load("queries.npz")
    for bit, kind in cases():
        x, init, d1, fr, keys = packed(orc, bit, kind)
        for i, slot in enumerate(G.QUERY_SLOTS):
            assert orc.simdselectd1(init, d1, bit, slot) == g["select"][bit, kind, 0, i]
            assert orc.simdselectFOR(init, fr, bit, slot & 127) == g["select"][bit, kind, 1, i]
        for ki, key in enumerate(keys):
            for li, length in enumerate(G.QUERY_LENGTHS):
                assert list(orc.simdsearchwithlengthd1(init, d1, bit, length, int(key))) == \
                    g["search"][bit, kind, 0, ki, li].tolist(), (bit, kind, key, length)
                assert list(orc.simdsearchwithlengthFOR(init, fr, bit, min(length, 127), int(key))) == \
                    g["search"][bit, kind, 1, ki, li].tolist(), (bit, kind, key, length)
            pos, val, o4 = orc.simdsearchd1(np.full(4, init, np.uint32), d1, bit, int(key))
            assert [pos, val] + [int(v) for v in o4] == g["searchd1"][bit, kind, ki].tolist(), (bit, kind, key)
        assert np.array_equal(orc.simdscand1(np.full(4, init, np.uint32), d1, bit), g["scan"][bit, kind])


def test_queries_against_reference(orc, ref):
    from simdcomp_amd import synth

    for bit in range(33):
        for seed in range(3):
            x = synth.rand_u32(40 * bit + seed, 128) if seed else synth.sorted_gaps(bit + 3, 128, min(bit, 22), first=7)
            init = 7 if not seed else int(synth.rand_u32(999 + seed, 1)[0])
            d1 = ref.simdpackwithoutmaskd1(init, x, bit)
            keys = np.concatenate([synth.rand_u32(5000 + bit, 6), x[[0, 17, 127]], [0, 0xFFFFFFFF]]).astype(np.uint32)
            o4 = np.array([3, 2, 1, init], np.uint32)
            for key in keys:
                for length in (128, 100, 5, 2, 1) + ((0,) if bit != 32 else ()):
                    assert ref.simdsearchwithlengthd1(init, d1, bit, length, int(key)) == \
                        orc.simdsearchwithlengthd1(init, d1, bit, length, int(key))
                a, b = ref.simdsearchd1(o4, d1, bit, int(key)), orc.simdsearchd1(o4, d1, bit, int(key))
                assert a[:2] == b[:2] and np.array_equal(a[2], b[2])
            for slot in range(0, 140, 7):
                assert ref.simdselectd1(init, d1, bit, slot) == orc.simdselectd1(init, d1, bit, slot)
            assert np.array_equal(ref.simdscand1(o4, d1, bit), orc.simdscand1(o4, d1, bit))
